# rocprofv3 kernel stats of the regression d = 4 bench (every task on the float64 path); run through gpurun
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_d4 -o d4 -- python bench.py --regression --d 4 --steps 5 --warmup 2 --no-cpu-baseline --no-parity --converge-steps 0 > gpurun_out/prof_d4.log 2>&1
python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_d4/d4_kernel_stats.csv')))
for r in rows[:8]:
    print(r['Name'][:80].ljust(80), r['Calls'], round(float(r['AverageNs'])/1000,1), r['Percentage'])
P
