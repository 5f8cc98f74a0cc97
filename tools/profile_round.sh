cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_final -o final -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/prof_final_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_final_fetch -o f -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity > gpurun_out/prof_final_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_final_write -o w -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity > gpurun_out/prof_final_write.log 2>&1
ls gpurun_out/prof_final gpurun_out/prof_final_fetch gpurun_out/prof_final_write
