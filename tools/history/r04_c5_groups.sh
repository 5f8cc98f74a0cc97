# round-4: task groups of the blocked path (ADKF_LG_GROUPS x ADKF_LG_SPREAD): tools/r04_c5_groups.sh T N ["g s" ...] (through gpurun)
# (historical: ADKF_LG_GROUPS / ADKF_LG_SPREAD exist at commit ef34797 only - the task-group experiment, profiles/r04_c5_groups.txt; on later trees
#  the variables are ignored and every line measures the same build)
cd $GRAFT_REPO_ROOT
T=$1; N=$2; shift 2
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-meta-test --converge-steps 0"
for cfg in "${@:-1 0}"; do
  set -- $cfg
  ADKF_LG_GROUPS=$1 ADKF_LG_SPREAD=$2 timeout -k 10 200 $B --tasks $T --n-support $N --n-query $N --d ${D:-512} 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('groups $1 spread $2 T=$T N=$N', d['ms_per_step'], d['value'])" || exit 1
done
