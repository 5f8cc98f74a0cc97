# round-2 profile set (run through gpurun): kernel stats of the default bench command + the two PMC passes for roofline.traffic
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r02 -o r02 -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --converge-steps 0 > gpurun_out/prof_r02_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_r02_fetch -o f -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --converge-steps 0 > gpurun_out/prof_r02_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_r02_write -o w -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --converge-steps 0 > gpurun_out/prof_r02_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/prof_r02_sq -o s -- python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --converge-steps 0 > gpurun_out/prof_r02_sq.log 2>&1
ls gpurun_out/prof_r02 gpurun_out/prof_r02_fetch gpurun_out/prof_r02_write gpurun_out/prof_r02_sq
