# A/B of the two-tasks-per-CU variant of the 128-point fit (run through gpurun): bench lines at 256 and 512 tasks, forced on / off
cd $GRAFT_REPO_ROOT
for T in 256 512; do for low in 0 1; do
  ADKF_INNER_LOWREG=$low python bench.py --tasks $T --steps 30 --warmup 5 --no-cpu-baseline --converge-steps 0 > gpurun_out/b_${T}_${low}.log 2>&1
  python - <<P
import json
j=json.loads(open('gpurun_out/b_${T}_${low}.log').read().strip().splitlines()[-1])
print('T=$T lowreg=$low: ms_per_step %.3f tasks/s %.0f k_inner %.3f ms' % (j['ms_per_step'], j['value'], j['roofline']['avg_launch_ms']), j.get('parity'))
P
done; done
