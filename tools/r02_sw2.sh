timeout -k 10 120 tools/sweepw_bench 256 2>&1 | tail -12
timeout -k 10 120 tools/sweepw_bench_stamp 256 2>&1 | grep -A9 "stamps of"
