#!/bin/bash
# the default bench line (all side objects), the C2 kernel stats and the T = 512 / T = 64 / Matern / C1 lines on the final tree (after the
# one-launch optimiser step).  Through gpurun; C3 / C5 / PMC evidence of tools/r05_round_end.sh is unchanged by that commit.
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time python bench.py ) > gpurun_out/r05_bench_c2.json 2> gpurun_out/r05_bench_c2.err || { tail -5 gpurun_out/r05_bench_c2.err; exit 1; }
tail -4 gpurun_out/r05_bench_c2.err
B="python bench.py --no-cpu-baseline --no-parity --no-meta-test --converge-steps 0 --side-configs off"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05 -o r05 -- $B --steps 20 --warmup 3 > gpurun_out/prof_r05_stats.log 2>&1 || exit 1
f=$(find gpurun_out/prof_r05 -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r05_bench_kernel_stats.csv; rm -rf gpurun_out/prof_r05 gpurun_out/prof_r05_stats.log
python bench.py --tasks 512 --steps 20 --warmup 5 --no-cpu-baseline --no-meta-test --side-configs off > gpurun_out/r05_bench_T512.json 2>/dev/null || exit 1
python bench.py --tasks 64 --steps 50 --warmup 10 --no-cpu-baseline --no-meta-test --side-configs off --no-parity --converge-steps 0 > gpurun_out/r05_bench_T64.json 2>/dev/null || exit 1
python bench.py --kernel matern --steps 20 --warmup 5 --no-cpu-baseline --no-meta-test --side-configs off > gpurun_out/r05_bench_c2_matern.json 2>/dev/null || exit 1
python bench.py --tasks 64 --n-support 32 --n-query 32 --d 64 --steps 50 --warmup 10 --no-cpu-baseline --no-meta-test --side-configs off > gpurun_out/r05_bench_c1.json 2>/dev/null || exit 1
for f in c2 T512 T64 c2_matern c1; do python -c "import json; d=json.loads(open('gpurun_out/r05_bench_$f.json').read().strip().splitlines()[-1]); print('$f', d['ms_per_step'], d['value'])"; done
