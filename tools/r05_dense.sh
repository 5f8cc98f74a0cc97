#!/bin/bash
# csrc/dense_x3.h through adkf_ift_amd/dense.py: its own tests, the extractor's GPU tests, then the C3 step with and without it.
mkdir -p gpurun_out
python -m pytest tests/test_gpu_dense.py tests/test_gpu_gnn.py -m gpu -x -q > gpurun_out/r05_dense_tests.log 2>&1; rc=$?
tail -6 gpurun_out/r05_dense_tests.log
[ $rc = 0 ] || exit 1
for m in 0 1 2 0 1 2; do
  export ADKF_X3_DENSE=$m ADKF_X3_DENSE_WGRAD=1
  if [ $m = 1 ]; then export ADKF_X3_DENSE_WGRAD=0; fi     # 1: forward and input gradient only; 2: the weight gradient as well
  if [ $m = 2 ]; then export ADKF_X3_DENSE=1; fi
  python tools/bench_c3.py --support 16 --query 128 --steps 4 --warmup 2 > gpurun_out/r05_c3_dense_$m.json 2> gpurun_out/r05_c3_dense_$m.err || { tail -5 gpurun_out/r05_c3_dense_$m.err; exit 1; }
  python -c "
import json; d=json.loads(open('gpurun_out/r05_c3_dense_$m.json').read().strip().splitlines()[-1]); print('dense=$m', {k: d[k] for k in d if 'ms' in k or 'tasks' in k})"
done
