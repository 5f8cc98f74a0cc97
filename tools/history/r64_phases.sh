# Phase times of the float64 path (k_tail64, one workgroup per flagged task) by early exit: ADKF_R64_STOP=k leaves after phase k
# (csrc/refine64.h, R64_STOP); every task of the d = 4 regression batch is flagged.  The duration of the FIRST k_tail64 launch of each
# run (later ones see the garbage the early exit leaves behind).  tools/r64_phases.sh (through gpurun) -> gpurun_out/r64_phases.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="python bench.py --regression --d 4 --tasks ${1:-64} --steps 2 --warmup 0 --no-cpu-baseline --no-parity --no-meta-test --converge-steps 0"
: > gpurun_out/r64_phases.txt
for k in 1 2 3 4 5 6 7 8 9 10 11 12 13 0; do
  rm -rf gpurun_out/prof_r64
  ADKF_R64_STOP=$k timeout -k 10 120 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/prof_r64 -o p -- $B > gpurun_out/prof_r64.log 2>&1
  python - $k >> gpurun_out/r64_phases.txt <<'P'
import csv, glob, sys
f = glob.glob('gpurun_out/prof_r64/**/*kernel_trace.csv', recursive=True)
rows = [r for r in csv.DictReader(open(f[0])) if 'k_tail64' in r['Kernel_Name']] if f else []
rows.sort(key=lambda r: int(r['Start_Timestamp']))
d = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3 for r in rows]
print(f"stop {sys.argv[1]:>2s}: first k_tail64 launch {d[0]:9.1f} us   (all: {' '.join('%.0f' % x for x in d[:4])})" if d else f"stop {sys.argv[1]}: no k_tail64 launch")
P
done
rm -rf gpurun_out/prof_r64
cat gpurun_out/r64_phases.txt
