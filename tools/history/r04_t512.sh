# round-4: the T=512 line (two tasks per CU) - kernel stats + both k_inner variants: tools/r04_t512.sh (through gpurun)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --tasks 512 --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-meta-test --converge-steps 0"
$B > gpurun_out/t512_default.json 2> gpurun_out/t512_default.err && tail -1 gpurun_out/t512_default.json | cut -c1-200 &&
ADKF_INNER_LOWREG=0 $B > gpurun_out/t512_resident.json 2> gpurun_out/t512_resident.err && tail -1 gpurun_out/t512_resident.json | cut -c1-200 &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_t512 -o t512 -- $B > gpurun_out/prof_t512.log 2>&1
echo "rocprof rc $?"
find gpurun_out/prof_t512 -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} gpurun_out/r04_t512_kernel_stats.csv
