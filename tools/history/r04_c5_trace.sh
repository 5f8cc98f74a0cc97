# round-4: kernel timeline of C5 with two task groups: tools/r04_c5_trace.sh (through gpurun)
# (historical: the task groups exist at commit ef34797 only, profiles/r04_c5_groups.txt; on later trees this traces the single-stream path)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export ADKF_LG_GROUPS=${1:-2} ADKF_LG_SPREAD=${2:-1}
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_c5g -o c5g -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-parity --no-meta-test --converge-steps 0 --tasks 8 --n-support 1024 --n-query 1024 --d 512 > gpurun_out/prof_c5g.log 2>&1
echo "rc $?"
python - <<'P'
import csv, glob
f = glob.glob('gpurun_out/prof_c5g/**/*kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
# last 40 % of the trace: steady state
n = len(rows); sel = rows[int(n * 0.6):int(n * 0.6) + 60]
t0 = int(sel[0]['Start_Timestamp'])
with open('gpurun_out/c5g_timeline.txt', 'w') as out:
    for r in sel:
        out.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} {(int(r['End_Timestamp']) - t0) / 1e3:9.1f} q{r.get('Queue_Id', '?')} {r['Kernel_Name'][:60]}\n")
print(open('gpurun_out/c5g_timeline.txt').read())
P
