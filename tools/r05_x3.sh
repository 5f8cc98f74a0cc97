#!/bin/bash
# gemm_x3.h (FP32 products on the BF16 matrix pipe): parity tests, then the C2 step with and without it.  Run through gpurun.
mkdir -p gpurun_out
what=${1:-all}
if [ "$what" = all ] || [ "$what" = tests ]; then
python -m pytest tests/test_gpu_parity.py tests/test_gpu_stress.py tests/test_gpu_reference_pins.py -m gpu -x -q > gpurun_out/r05_x3_tests.log 2>&1; rc=$?
tail -5 gpurun_out/r05_x3_tests.log
[ $rc = 0 ] || exit 1
fi
for m in 0 1 0 1; do
  export ADKF_X3=$m
  python bench.py --side-configs off --steps 300 --warmup 30 --no-cpu-baseline > gpurun_out/r05_x3_${m}.json 2> gpurun_out/r05_x3_${m}.err || { tail -5 gpurun_out/r05_x3_${m}.err; exit 1; }
  python - <<PY
import json
d=json.loads(open("gpurun_out/r05_x3_${m}.json").read().strip().splitlines()[-1])
print("x3=$m ms/step", round(d["ms_per_step"],4), "tasks/s", round(d["value"]), "parity", {k: (round(v,9) if isinstance(v,float) else v) for k,v in d.get("parity",{}).items()})
PY
done
