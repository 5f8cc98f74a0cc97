#!/bin/bash
# round 5: C5 (8 tasks, 1024 + 1024 points, d = 512) - the bench line, then rocprofv3 kernel stats of the same command.
# Usage (GPU box): bash tools/r05_c5_prof.sh [tag]
set -o pipefail
tag=${1:-c5}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--tasks 8 --n-support 1024 --n-query 1024 --d 512 --no-cpu-baseline --no-meta-test --side-configs off"
python bench.py --steps 10 --warmup 3 $ARGS > gpurun_out/r05_bench_$tag.json 2> gpurun_out/r05_bench_$tag.err || { tail -5 gpurun_out/r05_bench_$tag.err; exit 1; }
python -c "
import json; l=json.loads(open('gpurun_out/r05_bench_$tag.json').read().strip().splitlines()[-1])
print('C5', l['ms_per_step'], 'ms/step', l['value'], 'tasks/s; fit', l['roofline']['avg_launch_ms'], 'ms frac', l['roofline']['frac'], 'converged', l['converged'], 'parity', l['parity'])"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o c5 -- python bench.py --steps 5 --warmup 2 $ARGS --no-parity --converge-steps 0 > gpurun_out/prof_$tag.log 2>&1 || { tail -5 gpurun_out/prof_$tag.log; exit 1; }
f=$(find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/r05_${tag}_kernel_stats.csv
python - "$f" <<'P'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step', tot/7/1e6, 'launches per step', sum(int(r['Calls']) for r in rows)/7)
for r in rows[:26]:
    print(r['Name'][:90].ljust(90), int(r['Calls'])//7, round(float(r['AverageNs'])/1000,1), round(float(r['TotalDurationNs'])/7/1e6,3), r['Percentage'])
P
rm -rf gpurun_out/prof_$tag
