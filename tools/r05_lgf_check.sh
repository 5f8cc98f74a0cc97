#!/bin/bash
# round 5: the fused block step (csrc/large_fused.h) against the three-launch one: bit-identity + time per sweep (tools/lgf_bench.hip),
# then the library tests that go through the blocked path.  Usage (GPU box): bash tools/r05_lgf_check.sh
set -o pipefail
mkdir -p gpurun_out
L=gpurun_out/r05_lgf_bench.txt
: > $L
for args in "8 1024" "8 1000 1024" "3 300 300" "5 515 515" "16 1024" "64 256" "8 2048"; do
  echo "== lgf_bench $args" >> $L
  timeout -k 10 120 tools/lgf_bench $args >> $L 2>&1 || { echo "FAILED: $args" >> $L; tail -20 $L; exit 1; }
done
grep -E "==|differ|per sweep|FAILED" $L
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_block_step or blocked_path or c5_large or float64_path_beyond" --durations=5 > gpurun_out/r05_lgf_tests.log 2>&1 || { tail -30 gpurun_out/r05_lgf_tests.log; exit 1; }
tail -12 gpurun_out/r05_lgf_tests.log
