python -m pytest tests/test_gpu_stress.py -x -q -s -k "random_shapes or ill_conditioned" > gpurun_out/t_stress_r64.log 2>&1; echo "stress rc $?"
grep -a "FAIL\|worst\|passed\|failed" gpurun_out/t_stress_r64.log | cut -c1-250 | head -40
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-parity --converge-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('C2 ms/step', d['ms_per_step'], 'fit', d['roofline']['avg_launch_ms'])"
