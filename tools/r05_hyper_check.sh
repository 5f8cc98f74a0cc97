#!/bin/bash
# round 5: k_hyper after the register work (no scratch in any instance) - parity tests that go through it, then its launch time
# under rocprofv3 at C2 (RBF and Matern).  Usage (GPU box): bash tools/r05_hyper_check.sh
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_pins.py tests/test_gpu_stress.py tests/test_gpu_surface.py -m gpu -x -q --durations=8 > gpurun_out/r05_hyper_tests.log 2>&1 || { tail -30 gpurun_out/r05_hyper_tests.log; exit 1; }
tail -15 gpurun_out/r05_hyper_tests.log
for k in rbf matern; do
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r05_hyper_prof_$k -- python bench.py --steps 20 --warmup 5 --kernel $k --no-cpu-baseline --no-meta-test --no-parity --converge-steps 0 --side-configs off > gpurun_out/r05_hyper_bench_$k.json 2> gpurun_out/r05_hyper_bench_$k.err || { tail -5 gpurun_out/r05_hyper_bench_$k.err; exit 1; }
  f=$(find gpurun_out/r05_hyper_prof_$k -name '*kernel_stats.csv' | head -1)
  cp "$f" gpurun_out/r05_hyper_kernel_stats_$k.csv
  head -8 "$f" | cut -c1-150
  rm -rf gpurun_out/r05_hyper_prof_$k
done
