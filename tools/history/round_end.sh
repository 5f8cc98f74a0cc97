# end-of-round evidence (each line group fits one gpurun call of at most 1200 s):
#   1. python -m pytest tests -m gpu -x -q > gpurun_out/r02_gpu_tests.log      (9 min)
#   2. this script: bench lines at the BASELINE configurations               (5 min)
#   3. tools/round_end2.sh: strong-scaling rehearsal, C3, rocprofv3 passes   (4 min)
python bench.py > gpurun_out/r02_bench_c2.json 2> gpurun_out/r02_bench.err; echo "bench rc $?"
for T in 64 128 512; do python bench.py --steps 20 --warmup 5 --tasks $T --no-cpu-baseline --no-parity > gpurun_out/r02_bench_T$T.json 2>> gpurun_out/r02_bench.err; done
python bench.py --steps 20 --warmup 5 --tasks 64 --n-support 32 --n-query 32 --d 64 --no-cpu-baseline > gpurun_out/r02_bench_c1.json 2>> gpurun_out/r02_bench.err
python bench.py --steps 10 --warmup 3 --tasks 8 --n-support 1024 --n-query 1024 --d 512 --no-cpu-baseline --converge-steps 0 > gpurun_out/r02_bench_c5.json 2>> gpurun_out/r02_bench.err
ADKF_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-parity --converge-steps 0 > gpurun_out/r02_bench_gloo2.json 2>> gpurun_out/r02_bench.err; echo "gloo2 rc $?"



for f in c2 T64 T128 T512 c1 c5 gloo2; do echo "== $f"; tail -1 gpurun_out/r02_bench_$f.json | cut -c1-420; done
