# Round-3 evidence, part 1 (run through gpurun, ~6 min): bench lines of the BASELINE configurations into gpurun_out/r03_*.json
cd $GRAFT_REPO_ROOT
E=gpurun_out/r03_bench.err; : > $E
python bench.py > gpurun_out/r03_bench_c2.json 2>> $E; echo "c2 rc $?"
for T in 64 128 512; do python bench.py --steps 20 --warmup 5 --tasks $T --no-cpu-baseline --no-parity --converge-steps 0 > gpurun_out/r03_bench_T$T.json 2>> $E; done
ADKF_INNER_LOWREG=0 python bench.py --steps 20 --warmup 5 --tasks 512 --no-cpu-baseline --no-parity --converge-steps 0 > gpurun_out/r03_bench_T512_resident.json 2>> $E
python bench.py --steps 20 --warmup 5 --tasks 64 --n-support 32 --n-query 32 --d 64 --no-cpu-baseline > gpurun_out/r03_bench_c1.json 2>> $E
python bench.py --steps 10 --warmup 3 --tasks 8 --n-support 1024 --n-query 1024 --d 512 --no-cpu-baseline --converge-steps 2 > gpurun_out/r03_bench_c5.json 2>> $E
python bench.py --steps 5 --warmup 2 --tasks 64 --n-support 1024 --n-query 1024 --d 512 --no-cpu-baseline --no-parity --converge-steps 0 > gpurun_out/r03_bench_c5_T64.json 2>> $E
python bench.py --regression --steps 20 --warmup 5 --no-cpu-baseline --converge-steps 3 > gpurun_out/r03_bench_c2_regression.json 2>> $E
python bench.py --regression --d 4 --steps 10 --warmup 3 --no-cpu-baseline --converge-steps 0 > gpurun_out/r03_bench_regression_d4.json 2>> $E
python bench.py --d 4 --steps 10 --warmup 3 --no-cpu-baseline --converge-steps 0 > gpurun_out/r03_bench_classification_d4.json 2>> $E
ADKF_BENCH_BACKEND=gloo python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-parity --converge-steps 0 > gpurun_out/r03_bench_gloo2.json 2>> $E; echo "gloo2 rc $?"
for f in c2 T64 T128 T512 T512_resident c1 c5 c5_T64 c2_regression regression_d4 classification_d4 gloo2; do echo "== $f"; tail -1 gpurun_out/r03_bench_$f.json | cut -c1-300; done
tail -5 $E
