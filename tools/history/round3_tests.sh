# Round-3 evidence, part 3 (run through gpurun, ~6 min): the GPU test suite with its printed error tables, smoke(), and the float64-path lines
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -s --durations=10 > gpurun_out/r03_gpu_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r03_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03_smoke.log 2>&1; echo "smoke rc $?"
python bench.py --regression --d 4 --steps 10 --warmup 3 --no-cpu-baseline --converge-steps 0 > gpurun_out/r03_bench_regression_d4.json 2>> gpurun_out/r03_bench.err
python bench.py --d 4 --steps 10 --warmup 3 --no-cpu-baseline --converge-steps 0 > gpurun_out/r03_bench_classification_d4.json 2>> gpurun_out/r03_bench.err
for f in regression_d4 classification_d4; do tail -1 gpurun_out/r03_bench_$f.json | cut -c1-200; done
