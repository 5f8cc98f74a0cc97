# Round-3 evidence, part 2 (run through gpurun, ~6 min): rocprofv3 kernel stats of the default bench command, the PMC passes
# for roofline.traffic (FETCH_SIZE and WRITE_SIZE in SEPARATE passes, counters alone with --kernel-trace) and the SQ / LDS / MFMA
# counters of k_inner; the same for configuration 5.  tools/make_pmc_json.py turns them into profiles/r03_*.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-parity --converge-steps 0"
C5="--tasks 8 --n-support 1024 --n-query 1024 --d 512"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03 -o r03 -- $B --steps 20 --warmup 3 > gpurun_out/prof_r03_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_r03_fetch -o f -- $B --steps 5 --warmup 2 > gpurun_out/prof_r03_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_r03_write -o w -- $B --steps 5 --warmup 2 > gpurun_out/prof_r03_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/prof_r03_sq -o s -- $B --steps 5 --warmup 2 > gpurun_out/prof_r03_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/prof_r03_lds -o l -- $B --steps 5 --warmup 2 > gpurun_out/prof_r03_lds.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r03_c5 -o c5 -- $B $C5 --steps 5 --warmup 2 > gpurun_out/prof_r03_c5.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_r03_c5_fetch -o f -- $B $C5 --steps 3 --warmup 1 > gpurun_out/prof_r03_c5_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_r03_c5_write -o w -- $B $C5 --steps 3 --warmup 1 > gpurun_out/prof_r03_c5_write.log 2>&1
ls gpurun_out/prof_r03 gpurun_out/prof_r03_c5 | head; tail -2 gpurun_out/prof_r03_lds.log
