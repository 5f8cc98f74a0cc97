python -m pytest tests/test_gpu_stress.py -x -q -s -k "random_shapes" > gpurun_out/t_stress_d.log 2>&1; echo "stress rc $?"
grep -a "FAIL\|worst error\|passed\|failed" gpurun_out/t_stress_d.log | cut -c1-220 | head -30
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --converge-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 ms/step', d['ms_per_step'], 'fit', d['roofline']['avg_launch_ms'])"
