#!/bin/bash
# the default bench line (all side objects) and the C3 kernel stats on the final tree.  Through gpurun.
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
( time python bench.py ) > gpurun_out/r05_bench_c2.json 2> gpurun_out/r05_bench_c2.err || { tail -5 gpurun_out/r05_bench_c2.err; exit 1; }
tail -4 gpurun_out/r05_bench_c2.err
for shape in "16 128" "64 256"; do
  set -- $shape
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05_c3 -o c3 -- python tools/bench_c3.py --support $1 --query $2 --steps 3 --warmup 2 > gpurun_out/r05_bench_c3_$1_$2.json 2> gpurun_out/prof_r05_c3.err || { tail -3 gpurun_out/prof_r05_c3.err; exit 1; }
  f=$(find gpurun_out/prof_r05_c3 -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r05_c3_$1_$2_kernel_stats.csv; rm -rf gpurun_out/prof_r05_c3
done
rm -f gpurun_out/prof_r05_c3.err
B="python bench.py --no-cpu-baseline --no-parity --no-meta-test --converge-steps 0 --side-configs off"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05 -o r05 -- $B --steps 20 --warmup 3 > gpurun_out/prof_r05_stats.log 2>&1 || exit 1
f=$(find gpurun_out/prof_r05 -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r05_bench_kernel_stats.csv; rm -rf gpurun_out/prof_r05 gpurun_out/prof_r05_stats.log
python bench.py --tasks 512 --steps 20 --warmup 5 --no-cpu-baseline --no-meta-test --side-configs off > gpurun_out/r05_bench_T512.json 2>/dev/null || exit 1
python bench.py --kernel matern --steps 20 --warmup 5 --no-cpu-baseline --no-meta-test --side-configs off > gpurun_out/r05_bench_c2_matern.json 2>/dev/null || exit 1
python bench.py --tasks 64 --n-support 32 --n-query 32 --d 64 --steps 50 --warmup 10 --no-cpu-baseline --no-meta-test --side-configs off > gpurun_out/r05_bench_c1.json 2>/dev/null || exit 1
