cd $GRAFT_REPO_ROOT
python bench.py --regression --d 4 --steps 10 --warmup 3 --no-cpu-baseline --converge-steps 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('reg d4 ms', j['ms_per_step'], j['parity'])"
python bench.py --d 4 --steps 10 --warmup 3 --no-cpu-baseline --converge-steps 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cls d4 ms', j['ms_per_step'], j['parity'])"
python -m pytest tests/test_gpu_stress.py -m gpu -q 2>&1 | tail -3
