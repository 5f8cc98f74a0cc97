#!/bin/bash
# round 5 experiment: the 128 x 128 output tile of k_bgemm for the N^3-sized products of the outer stage at the C5 shape
set -o pipefail
mkdir -p gpurun_out
ARGS="--tasks 8 --n-support 1024 --n-query 1024 --d 512 --no-cpu-baseline --no-meta-test --side-configs off --converge-steps 0 --steps 10 --warmup 3"
for m in 0 512; do
  ADKF_GEMM_TILE128_MIN=$m python bench.py $ARGS > gpurun_out/r05_c5_tile128_$m.json 2> gpurun_out/r05_c5_tile128_$m.err || { tail -5 gpurun_out/r05_c5_tile128_$m.err; exit 1; }
  python -c "
import json; l=json.loads(open('gpurun_out/r05_c5_tile128_$m.json').read().strip().splitlines()[-1])
print('TILE128_MIN=$m', round(l['ms_per_step'],3), 'ms/step, fit', round(l['roofline']['avg_launch_ms'],3), 'parity', l['parity'])"
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ADKF_GEMM_TILE128_MIN=512 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t128 -o c5 -- python bench.py $ARGS --no-parity > gpurun_out/prof_t128.log 2>&1 || exit 1
f=$(find gpurun_out/prof_t128 -name '*kernel_stats.csv' | head -1)
grep -E "k_bgemm<adkf::Prob(C|S|P|OC|MA|Mixed|DZ|Cres|Cfix)" "$f" | cut -d, -f1,2,4 | cut -c1-120
rm -rf gpurun_out/prof_t128
ADKF_GEMM_TILE128_MIN=512 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "c5_large or blocked_path" 2>&1 | tail -3
