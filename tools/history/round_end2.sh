# second half of tools/round_end.sh (one gpurun call may run 1200 s): strong-scaling rehearsal, C3, profiles
ADKF_BENCH_BACKEND=gloo python bench.py --gpus 2 --global-tasks 512 --steps 10 --warmup 3 --no-cpu-baseline --no-parity --converge-steps 0 > gpurun_out/r02_bench_gloo2_strong.json 2>> gpurun_out/r02_bench.err
python tools/bench_c3.py --steps 5 --warmup 2 > gpurun_out/r02_bench_c3.json 2>> gpurun_out/r02_bench.err
python tools/bench_c3.py --steps 5 --warmup 2 --gemm-tuning off > gpurun_out/r02_bench_c3_untuned.json 2>> gpurun_out/r02_bench.err
bash tools/profile_round.sh > /dev/null 2>&1
for f in gloo2_strong c3 c3_untuned; do echo "== $f"; tail -1 gpurun_out/r02_bench_$f.json | cut -c1-420; done
