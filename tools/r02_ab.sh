for i in 1 2; do
ADKF_LIB=$PWD/tools/libadkf_blk.so python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --converge-steps 0 > gpurun_out/ab_blk_$i.json 2>/dev/null
python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-parity --converge-steps 0 > gpurun_out/ab_w_$i.json 2>/dev/null
done
python - <<'PY'
import json
for f in ["ab_blk_1","ab_w_1","ab_blk_2","ab_w_2"]:
    d=json.loads(open(f"gpurun_out/{f}.json").read().strip().splitlines()[-1])
    print(f, "ms/step %.3f" % d["ms_per_step"], "fit ms %.3f" % d["roofline"]["avg_launch_ms"])
PY
python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -3
