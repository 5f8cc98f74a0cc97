#!/bin/bash
# A/B of the streaming dense forms on ONE box: the C2 step with ADKF_DENSE_STREAM=0 / 1 alternating.  Through gpurun.
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-meta-test --no-parity --side-configs off --converge-steps 0"
for rep in 1 2 3; do
  for v in 0 1; do
    ADKF_DENSE_STREAM=$v timeout -k 10 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ADKF_DENSE_STREAM=$v', round(d['ms_per_step'],4), 'ms', round(d['value']), 'tasks/s')" || exit 1
  done
done
