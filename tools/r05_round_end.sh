#!/bin/bash
# Round-5 evidence, two gpurun calls:
#   bash tools/r05_round_end.sh tests            the whole GPU suite + smoke()
#   bash tools/r05_round_end.sh profiles <commit>   the default bench line (every BASELINE config as side objects), rocprofv3 kernel stats + PMC
#                                                passes of the C2 command and of C5, the ARD / float64-path / T = 512 lines, C3 kernel stats
# Summaries land in gpurun_out/r05_* (copy them to profiles/ afterwards); raw rocprofv3 directories are deleted on the box.
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
if [ "$1" = tests ]; then
  python -m pytest tests -m gpu -x -q -s --durations=12 > gpurun_out/r05_gpu_tests.log 2>&1; rc=$?
  grep -E "passed|failed|error" gpurun_out/r05_gpu_tests.log | tail -3
  [ $rc -ne 0 ] && { tail -40 gpurun_out/r05_gpu_tests.log; exit $rc; }
  python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r05_smoke.log 2>&1 || { tail -5 gpurun_out/r05_smoke.log; exit 1; }
  tail -1 gpurun_out/r05_smoke.log
  exit 0
fi
commit=${2:-unknown}
( time python bench.py ) > gpurun_out/r05_bench_c2.json 2> gpurun_out/r05_bench_c2.err || { tail -5 gpurun_out/r05_bench_c2.err; exit 1; }
tail -4 gpurun_out/r05_bench_c2.err
B="python bench.py --no-cpu-baseline --no-parity --no-meta-test --converge-steps 0 --side-configs off"
C5="--tasks 8 --n-support 1024 --n-query 1024 --d 512"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05 -o r05 -- $B --steps 20 --warmup 3 > gpurun_out/prof_r05_stats.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_r05_fetch -o f -- $B --steps 5 --warmup 2 > gpurun_out/prof_r05_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_r05_write -o w -- $B --steps 5 --warmup 2 > gpurun_out/prof_r05_write.log 2>&1 || exit 1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/prof_r05_sq -o s -- $B --steps 5 --warmup 2 > gpurun_out/prof_r05_sq.log 2>&1 || exit 1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/prof_r05_lds -o l -- $B --steps 5 --warmup 2 > gpurun_out/prof_r05_lds.log 2>&1 || exit 1
echo "C2 profiles done"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05_c5 -o c5 -- $B $C5 --steps 5 --warmup 2 > gpurun_out/prof_r05_c5.log 2>&1 || exit 1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_r05_c5_fetch -o f -- $B $C5 --steps 3 --warmup 1 > gpurun_out/prof_r05_c5_fetch.log 2>&1 || exit 1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_r05_c5_write -o w -- $B $C5 --steps 3 --warmup 1 > gpurun_out/prof_r05_c5_write.log 2>&1 || exit 1
echo "C5 profiles done"
python tools/make_pmc_json.py r05 $commit > gpurun_out/r05_pmc_summary.log 2>&1 || { tail -5 gpurun_out/r05_pmc_summary.log; exit 1; }
rm -f gpurun_out/prof_r05*.log
python bench.py --steps 10 --warmup 3 $C5 --no-cpu-baseline --no-meta-test --side-configs off > gpurun_out/r05_bench_c5.json 2> gpurun_out/r05_bench_c5.err || exit 1
python bench.py --tasks 512 --steps 20 --warmup 5 --no-cpu-baseline --no-meta-test --side-configs off > gpurun_out/r05_bench_T512.json 2>/dev/null || exit 1
python bench.py --tasks 64 --steps 50 --warmup 10 --no-cpu-baseline --no-meta-test --side-configs off --no-parity --converge-steps 0 > gpurun_out/r05_bench_T64.json 2>/dev/null || exit 1
python bench.py --kernel matern --steps 20 --warmup 5 --no-cpu-baseline --no-meta-test --side-configs off > gpurun_out/r05_bench_c2_matern.json 2>/dev/null || exit 1
python bench.py --regression --d 4 --steps 10 --warmup 3 --no-cpu-baseline --no-meta-test --side-configs off > gpurun_out/r05_bench_regression_d4.json 2>/dev/null || exit 1
python bench.py --d 4 --steps 10 --warmup 3 --no-cpu-baseline --no-meta-test --side-configs off > gpurun_out/r05_bench_classification_d4.json 2>/dev/null || exit 1
python bench.py --ard --steps 10 --warmup 3 --no-cpu-baseline --no-meta-test --side-configs off > gpurun_out/r05_bench_ard.json 2>/dev/null || exit 1
echo "bench lines done"
for shape in "16 128" "64 256"; do
  set -- $shape
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r05_c3 -o c3 -- python tools/bench_c3.py --support $1 --query $2 --steps 3 --warmup 2 > gpurun_out/r05_bench_c3_$1_$2.json 2> gpurun_out/prof_r05_c3.err || { tail -3 gpurun_out/prof_r05_c3.err; exit 1; }
  f=$(find gpurun_out/prof_r05_c3 -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r05_c3_$1_$2_kernel_stats.csv; rm -rf gpurun_out/prof_r05_c3
done
rm -f gpurun_out/prof_r05_c3.err
ls gpurun_out | grep r05_ | head -40
