# A/B helper (run through gpurun): parity subset, bench line, kernel stats into gpurun_out/prof_x
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_parity.py tests/test_gpu_surface.py tests/test_gpu_reference_pins.py -x -q -m gpu > gpurun_out/t.log 2>&1; tail -3 gpurun_out/t.log
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --converge-steps 0 > gpurun_out/b.log 2>&1 && python - <<'P'
import json
j=json.loads(open('gpurun_out/b.log').read().strip().splitlines()[-1])
print('ms_per_step', j['ms_per_step'], 'k_inner', j['roofline']['avg_launch_ms'], j['parity'])
P
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_x -o x -- python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --converge-steps 0 > gpurun_out/prof_x.log 2>&1
python - <<'P'
import csv
for r in csv.DictReader(open('gpurun_out/prof_x/x_kernel_stats.csv')):
    if float(r['Percentage'])>1.0: print(r['Name'][:70].ljust(70), r['Calls'], round(float(r['AverageNs'])/1000,1))
P
