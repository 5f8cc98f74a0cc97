# round-4: C3 (default GNN+ECFP+fc model, 16 tasks x (16 + 128) molecules): step time and kernel stats.  tools/r04_c3.sh <tag>
TAG=${1:-r04_c3}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
python tools/bench_c3.py --steps 5 > gpurun_out/${TAG}.json 2> gpurun_out/${TAG}.err; echo "bench_c3 rc $?"; cat gpurun_out/${TAG}.json
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python tools/bench_c3.py --steps 3 --warmup 2 > gpurun_out/prof_${TAG}.log 2>&1
echo "rocprof rc $?"
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_kernel_stats.csv
rm -f gpurun_out/prof_$TAG/*kernel_trace.csv
