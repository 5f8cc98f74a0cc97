# Round-4 evidence for configuration 5 (through gpurun, ~3 min): kernel stats and the two PMC passes behind roofline.traffic of the
# blocked fit; tools/make_pmc_json.py c5 turns them into profiles/r04_c5_kernel_stats.csv and profiles/r04_c5_fit_pmc.json.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-parity --no-meta-test --converge-steps 0 --tasks 8 --n-support 1024 --n-query 1024 --d 512"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04_c5b -o c5 -- $B --steps 5 --warmup 2 > gpurun_out/prof_r04_c5b.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_r04_c5_fetch -o f -- $B --steps 3 --warmup 1 > gpurun_out/prof_r04_c5_fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_r04_c5_write -o w -- $B --steps 3 --warmup 1 > gpurun_out/prof_r04_c5_write.log 2>&1
echo "rc $?"
find gpurun_out/prof_r04_c5* -name "*kernel_trace.csv" -size +20M -delete
tail -1 gpurun_out/prof_r04_c5b.log | cut -c1-300
