#!/bin/bash
# round 5, mid-round check: blocked-path tests + C5 line; the small shapes through the one-kernel outer stage (golden / stress / pins /
# surface tests, C1 line with and without it); the default-width extractor tests with their fixed bounds
set -o pipefail
mkdir -p gpurun_out
bash tools/r05_lg_tests.sh c5_fused4 || exit 1
python -m pytest tests/test_gpu_parity.py tests/test_gpu_reference_pins.py tests/test_gpu_stress.py tests/test_gpu_surface.py tests/test_gpu_determinism.py -m gpu -x -q -k "not blocked_path and not c5_large and not fused_block_step and not float64_path_beyond and not two_tasks_per_cu" > gpurun_out/r05_small_tests.log 2>&1 || { tail -30 gpurun_out/r05_small_tests.log; exit 1; }
tail -3 gpurun_out/r05_small_tests.log
C1="--tasks 64 --n-support 32 --n-query 32 --d 64 --steps 200 --warmup 20 --no-cpu-baseline --no-meta-test --side-configs off --no-parity --converge-steps 0"
python bench.py $C1 > gpurun_out/r05_bench_c1.json 2> gpurun_out/r05_bench_c1.err || { tail -5 gpurun_out/r05_bench_c1.err; exit 1; }
ADKF_FUSED_OUTER_MIN=65 python bench.py $C1 > gpurun_out/r05_bench_c1_sixteen_launches.json 2> gpurun_out/r05_bench_c1_16.err || exit 1
python bench.py $C1 --graph > gpurun_out/r05_bench_c1_graph.json 2> gpurun_out/r05_bench_c1_graph.err || exit 1
python - <<'P'
import json
for f in ("r05_bench_c1", "r05_bench_c1_sixteen_launches", "r05_bench_c1_graph"):
    l = json.loads(open("gpurun_out/%s.json" % f).read().strip().splitlines()[-1])
    print(f, round(l["ms_per_step"], 4), "ms/step", round(l["value"]), "tasks/s; host enqueue", round(l["host_enqueue_ms_per_step"], 4))
P
python -m pytest tests/test_gpu_gnn.py -m gpu -x -q -s -k "default_width or c3_default or fused_block_stage" > gpurun_out/r05_gnn_tests.log 2>&1 || { tail -30 gpurun_out/r05_gnn_tests.log; exit 1; }
grep -E "default-width|C3 default|fused vs unfused|passed|failed" gpurun_out/r05_gnn_tests.log
