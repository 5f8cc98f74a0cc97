# Round-4 evidence (run through gpurun, ~5 min): rocprofv3 kernel stats of the default bench command, the PMC passes for
# roofline.traffic (FETCH_SIZE and WRITE_SIZE in SEPARATE passes, counters alone with --kernel-trace) and the SQ / LDS / MFMA
# counters of k_inner and k_hyper; kernel stats of configuration 5.  tools/make_pmc_json.py r04 turns them into profiles/r04_*.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --no-cpu-baseline --no-parity --no-meta-test --converge-steps 0"
C5="--tasks 8 --n-support 1024 --n-query 1024 --d 512"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04 -o r04 -- $B --steps 20 --warmup 3 > gpurun_out/prof_r04_stats.log 2>&1
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_r04_fetch -o f -- $B --steps 5 --warmup 2 > gpurun_out/prof_r04_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/prof_r04_write -o w -- $B --steps 5 --warmup 2 > gpurun_out/prof_r04_write.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d gpurun_out/prof_r04_sq -o s -- $B --steps 5 --warmup 2 > gpurun_out/prof_r04_sq.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES --kernel-trace --output-format csv -d gpurun_out/prof_r04_lds -o l -- $B --steps 5 --warmup 2 > gpurun_out/prof_r04_lds.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_r04_c5 -o c5 -- $B $C5 --steps 5 --warmup 2 > gpurun_out/prof_r04_c5.log 2>&1
find gpurun_out/prof_r04* -name "*kernel_trace.csv" -size +20M -delete
ls gpurun_out/prof_r04 | head -3; tail -1 gpurun_out/prof_r04_lds.log | cut -c1-200
