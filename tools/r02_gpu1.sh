# round-2 first GPU pass: new parity tests, then bench lines at the C2 shape and the per-rank shapes of C4
set -x
python -m pytest tests/test_gpu_reference_pins.py tests/test_gpu_gnn.py -x -q -s > gpurun_out/t_pins.log 2>&1; echo "pins rc $?" 
python -m pytest tests/test_gpu_parity.py -x -q -s > gpurun_out/t_parity.log 2>&1; echo "parity rc $?"
python bench.py --steps 20 --warmup 5 > gpurun_out/b_c2.json 2> gpurun_out/b_c2.err; echo "bench rc $?"
for T in 64 128 512; do python bench.py --steps 20 --warmup 5 --tasks $T --no-cpu-baseline --no-parity > gpurun_out/b_T$T.json 2>> gpurun_out/b_c2.err; done
tail -3 gpurun_out/t_pins.log gpurun_out/t_parity.log
cat gpurun_out/b_c2.json gpurun_out/b_T64.json gpurun_out/b_T128.json gpurun_out/b_T512.json | cut -c1-600
