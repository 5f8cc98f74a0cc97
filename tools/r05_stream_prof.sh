#!/bin/bash
# kernel stats of the C2 step with the stand-in feature map's forward product on the library GEMM (0) / on k_dense3_sk (1), on one box.
# Through gpurun.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-meta-test --no-parity --side-configs off --converge-steps 0"
for v in 0 1; do
  export ADKF_X3_FWD=$v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_s3_$v -o s3 -- $B > gpurun_out/prof_s3_$v.log 2>&1 || { tail -5 gpurun_out/prof_s3_$v.log; exit 1; }
  f=$(find gpurun_out/prof_s3_$v -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/s3_fwd${v}_kernel_stats.csv; rm -rf gpurun_out/prof_s3_$v gpurun_out/prof_s3_$v.log
done
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-meta-test --no-parity --side-configs off --converge-steps 0"
for rep in 1 2 3; do
  for v in 0 1; do
    ADKF_X3_FWD=$v timeout -k 10 200 $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readline()); print('ADKF_X3_FWD=$v', round(d['ms_per_step'],4), 'ms', round(d['value']), 'tasks/s')" || exit 1
  done
done
