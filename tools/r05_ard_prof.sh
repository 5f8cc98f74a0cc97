#!/bin/bash
# round 5: the ARD path (h = 2 + d = 258 inner parameters at the C2 shape) - bench line + rocprofv3 kernel stats
set -o pipefail
tag=${1:-ard}
mkdir -p gpurun_out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--ard --no-cpu-baseline --no-meta-test --side-configs off"
python bench.py --steps 10 --warmup 3 $ARGS > gpurun_out/r05_bench_$tag.json 2> gpurun_out/r05_bench_$tag.err || { tail -5 gpurun_out/r05_bench_$tag.err; exit 1; }
python -c "
import json; l=json.loads(open('gpurun_out/r05_bench_$tag.json').read().strip().splitlines()[-1])
print('ARD', l['ms_per_step'], 'ms/step', l['value'], 'tasks/s; parity', l['parity'], l.get('ard'))"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -o ard -- python bench.py --steps 5 --warmup 2 $ARGS --no-parity > gpurun_out/prof_$tag.log 2>&1 || { tail -5 gpurun_out/prof_$tag.log; exit 1; }
f=$(find gpurun_out/prof_$tag -name '*kernel_stats.csv' | head -1)
cp "$f" gpurun_out/r05_${tag}_kernel_stats.csv
python - "$f" <<'P'
import csv, sys
rows=list(csv.DictReader(open(sys.argv[1])))
tot=sum(float(r['TotalDurationNs']) for r in rows)
print('total kernel ms per step', tot/7/1e6, 'launches per step', sum(int(r['Calls']) for r in rows)/7)
for r in rows[:24]:
    print(r['Name'][:90].ljust(90), round(int(r['Calls'])/7,1), round(float(r['AverageNs'])/1000,1), round(float(r['TotalDurationNs'])/7/1e6,3), r['Percentage'])
P
rm -rf gpurun_out/prof_$tag
