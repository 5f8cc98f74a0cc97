# Round-3 (second session) check of the tree: GPU test suite, smoke(), the C2 bench line (run through gpurun, ~8 min)
cd $GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x --durations=10 > gpurun_out/r03b_gpu_tests.log 2>&1; echo "pytest rc $?"; tail -3 gpurun_out/r03b_gpu_tests.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r03b_smoke.log 2>&1; echo "smoke rc $?"
python bench.py > gpurun_out/r03b_bench_c2.json 2> gpurun_out/r03b_bench.err; echo "bench rc $?"; tail -1 gpurun_out/r03b_bench_c2.json | cut -c1-400
