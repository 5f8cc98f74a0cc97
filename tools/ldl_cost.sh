cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for th in 0 1e30; do
  ADKF_LDL_THRESHOLD=$th python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>&1 | tail -1 | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('threshold $th:', d['ms_per_step'], 'ms/step', d['parity'])"
done
ADKF_LDL_THRESHOLD=0 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_ldl -o ldl -- python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-parity > gpurun_out/prof_ldl.log 2>&1
grep -i "k_ldl" gpurun_out/prof_ldl/ldl_kernel_stats.csv | cut -c1-120
