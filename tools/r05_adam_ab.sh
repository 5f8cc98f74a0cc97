#!/bin/bash
# A/B on one box: the optimiser's two launches (+ the split launch of the next forward) against adkf_clip_adam_step_one
F="--steps 300 --warmup 30 --no-cpu-baseline --no-parity --no-meta-test --side-configs off --converge-steps 0"
for r in 1 2; do
  for v in 0 1; do
    ADKF_CLIP_ADAM_ONE=$v python bench.py $F 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('ADKF_CLIP_ADAM_ONE=$v', d['ms_per_step'], d['value'])"
  done
done
