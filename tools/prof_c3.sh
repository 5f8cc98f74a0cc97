cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_c3 -o c3 -- python tools/bench_c3.py --steps 5 --warmup 2 > gpurun_out/prof_c3.log 2>&1
tail -2 gpurun_out/prof_c3.log
python - <<'P'
import csv
rows=list(csv.DictReader(open('gpurun_out/prof_c3/c3_kernel_stats.csv')))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step', tot/7/1e6)
for r in rows[:32]:
    print(r['Name'][:110].ljust(110), r['Calls'], round(float(r['TotalDurationNs'])/7/1e6,3), r['Percentage'])
P
