#!/bin/bash
# round 5: the fused block step (csrc/large_fused.h) against the three-launch one: bit-identity + time per sweep and per launch
# (tools/lgf_bench.hip; a fourth argument = extra dynamic LDS, 40000 leaves one workgroup per CU), then the library tests that go through
# the blocked path.  Usage (GPU box): bash tools/r05_lgf_check.sh [quick]
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r05_lgf_bench.txt
: > $L
if [ "$1" = quick ]; then SHAPES=("8 1024 1024 0 0 0" "8 1024 1024 0 1 0" "8 1024 1024 0 2 0" "8 1000 1024 0 0 0" "16 1024 1024 0 0 0" "64 256 256 0 0 0" "3 300 300 0 0 0" "5 515 515 0 0 0"); else SHAPES=("8 1024" "8 1000 1024" "3 300 300" "5 515 515" "16 1024" "64 256" "8 2048"); fi
for args in "${SHAPES[@]}"; do
  echo "== lgf_bench $args" >> $L
  timeout -k 10 120 tools/lgf_bench $args >> $L 2>&1 || { echo "FAILED: $args" >> $L; tail -20 $L; exit 1; }
done
grep -E "==|differ|per sweep|per launch|FAILED" $L | grep -v "fused run [1-4]"
[ "$1" = quick ] && exit 0
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_block_step or blocked_path or c5_large or float64_path_beyond" --durations=5 > gpurun_out/r05_lgf_tests.log 2>&1 || { tail -30 gpurun_out/r05_lgf_tests.log; exit 1; }
tail -12 gpurun_out/r05_lgf_tests.log
