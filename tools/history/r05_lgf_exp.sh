for e in 0 1 2 3 4; do echo "== EXP $e"; timeout -k 10 120 tools/lgf_bench_exp$e 8 1024 1024 0 0 0 2>&1 | grep -E "fused\)|look = 0|k_lg_diag|per sweep"; done
