timeout -k 10 120 tools/sweepw_bench_stamp 256 > gpurun_out/sweepw_stamp.log 2>&1; echo "stamp rc $?"
grep -A9 "stamps of" gpurun_out/sweepw_stamp.log; tail -3 gpurun_out/sweepw_stamp.log
timeout -k 10 600 python tools/diag_stress.py 7 20 42 44 51 > gpurun_out/diag_stress.log 2>&1; echo "diag rc $?"
cat gpurun_out/diag_stress.log | cut -c1-250
