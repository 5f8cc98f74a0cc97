# Round-3 evidence, part 4 (run through gpurun, ~3 min): configuration 3 (default GNN + ECFP + fc model, 16 tasks of 16 + 128 molecules)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/bench_c3.py --steps 5 --warmup 2 > gpurun_out/r03_bench_c3.json 2> gpurun_out/r03_bench_c3.err; echo "c3 rc $?"
python tools/bench_c3.py --steps 5 --warmup 2 --gemm-tuning off > gpurun_out/r03_bench_c3_untuned.json 2>> gpurun_out/r03_bench_c3.err
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_c3 -o c3 -- python tools/bench_c3.py --steps 5 --warmup 2 > gpurun_out/prof_r03_c3.log 2>&1
for f in c3 c3_untuned; do tail -1 gpurun_out/r03_bench_$f.json | cut -c1-400; done
