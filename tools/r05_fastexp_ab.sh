#!/bin/bash
# k_hyper with ONE v_exp_f32 per exponential factor (-DADKF_HY_FAST_EXP=1, built as tools/variants/libadkf_gp_fastexp.so) against libm's
# expf: the tests that go through it (-s: their error / tolerance lines), then its launch time at C2 (RBF and Matern) for both builds.
set -o pipefail
mkdir -p gpurun_out
export TMPDIR=/tmp
V=$GRAFT_REPO_ROOT/tools/variants/libadkf_gp_fastexp.so
ADKF_LIB=$V python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_pins.py tests/test_gpu_stress.py tests/test_gpu_properties.py -m gpu -x -q -s > gpurun_out/s4_fastexp_tests.log 2>&1; rc=$?
grep -E "worst|passed|failed" gpurun_out/s4_fastexp_tests.log | cut -c1-400
[ $rc -ne 0 ] && { tail -30 gpurun_out/s4_fastexp_tests.log; exit $rc; }
for k in rbf matern; do
  for v in libm fast; do
    lib=""; [ $v = fast ] && lib=$V
    ADKF_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/s4_fx_prof -- python3 bench.py --steps 30 --warmup 5 --kernel $k --no-cpu-baseline --no-meta-test --no-parity --converge-steps 0 --side-configs off > gpurun_out/s4_fx_bench.json 2> gpurun_out/s4_fx_bench.err || { tail -5 gpurun_out/s4_fx_bench.err; exit 1; }
    f=$(find gpurun_out/s4_fx_prof -name '*kernel_stats.csv' | head -1)
    echo "$k $v: $(grep k_hyper "$f" | cut -d, -f1-4 | cut -c1-120) step $(python3 -c "import json;print(json.loads(open('gpurun_out/s4_fx_bench.json').read().strip().splitlines()[-1])['ms_per_step'])")"
    rm -rf gpurun_out/s4_fx_prof
  done
done
