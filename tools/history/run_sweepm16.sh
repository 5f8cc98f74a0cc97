#!/bin/bash
# runs every tools/sweepm_bench_16* binary (the sixteen-pivot sweep and its ablation / stamp builds), prints correctness of the
# plain build and the timing lines (gpurun)
for b in tools/sweepm_bench_m4 tools/sweepm_bench_16*; do
  [ -x "$b" ] || continue
  echo "== $b"
  if [ "$b" = tools/sweepm_bench_16 ]; then timeout -k 10 60 ./$b 256 2>&1 | grep -v "^blocked"
  else timeout -k 10 60 ./$b 256 2>&1 | grep -E "us per sweep|wave [0-9]:|stamps" | grep -v "^blocked"; fi
done
