# rocprofv3 kernel stats of configuration 5 (8 tasks, 1024 + 1024 points, d = 512: the blocked path); run through gpurun
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c5 -o c5 -- python bench.py --steps 5 --warmup 2 --tasks 8 --n-support 1024 --n-query 1024 --d 512 --no-cpu-baseline --no-parity --converge-steps 0 > gpurun_out/prof_c5.log 2>&1
python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_c5/c5_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step', tot/7/1e6)
for r in rows[:22]:
    print(r['Name'][:100].ljust(100), int(r['Calls'])//7, round(float(r['AverageNs'])/1000,1), round(float(r['TotalDurationNs'])/7/1e6,3), r['Percentage'])
P
