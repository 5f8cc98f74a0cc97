timeout -k 10 120 tools/sweepw_bench 256 2>&1 | tail -10
for v in 24 31; do echo "== ablate $v"; timeout -k 10 120 tools/sweepw_abl_$v 256 2>&1 | grep "wave-owned: "; done
