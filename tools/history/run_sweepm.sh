#!/bin/bash
# runs every tools/sweepm_bench* binary (+ the VALU variant as the box's yardstick), prints the timing lines (gpurun)
for b in tools/sweepw_bench tools/sweepm_bench tools/sweepm_bench_*; do
  [ -x "$b" ] || continue
  echo "== $b"; timeout -k 10 60 ./$b 256 2>&1 | grep -E "us per sweep|wave [0-9]:|stamps|FAILED|OK" | grep -v "^blocked"
done
