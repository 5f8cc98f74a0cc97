# round-4: kernel stats of the default bench command (run through gpurun): tools/r04_stats.sh <tag>
TAG=${1:-r04}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$TAG -o $TAG -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-meta-test --converge-steps 0 > gpurun_out/prof_${TAG}_stats.log 2>&1
echo "rocprof rc $?"
find gpurun_out/prof_$TAG -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/${TAG}_kernel_stats.csv
tail -1 gpurun_out/prof_${TAG}_stats.log | cut -c1-300
