cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ard -o ard -- python bench.py --ard --steps 10 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/prof_ard.log 2>&1
tail -1 gpurun_out/prof_ard.log | cut -c1-300
