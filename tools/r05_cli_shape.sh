#!/bin/bash
# the GP section alone at the reference's CLI training shape (support 64, query 256, Matern-5/2: fs_mol/adaptive_dkt_train.py:50-61,112)
# with the synthetic feature map: 16 tasks per step (the CLI batch) and 256 (one per CU); kernel stats of both.  Through gpurun.
set -o pipefail
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
S="--n-support 64 --n-query 256 --d 256 --kernel matern --no-cpu-baseline --no-meta-test --side-configs off"
for T in 16 256; do
  python bench.py --tasks $T $S --steps 30 --warmup 5 > gpurun_out/r05_bench_cli_shape_T$T.json 2> gpurun_out/r05_cli.err || { tail -5 gpurun_out/r05_cli.err; exit 1; }
  python -c "import json; d=json.loads(open('gpurun_out/r05_bench_cli_shape_T$T.json').read().strip().splitlines()[-1]); print('T=$T', d['ms_per_step'], d['value'], d.get('host_enqueue_ms_per_step'), d.get('parity'), d.get('converged'))"
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cli -o cli -- python bench.py --tasks $T $S --no-parity --converge-steps 0 --steps 20 --warmup 3 > gpurun_out/prof_cli.log 2>&1 || exit 1
  f=$(find gpurun_out/prof_cli -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r05_cli_shape_T${T}_kernel_stats.csv; rm -rf gpurun_out/prof_cli gpurun_out/prof_cli.log
done
rm -f gpurun_out/r05_cli.err
