timeout -k 10 120 tools/sweepw_bench 256 2>&1 | tail -20
timeout -k 10 120 tools/sweepw_abl_8 256 2>&1 | grep "wave-owned: "
