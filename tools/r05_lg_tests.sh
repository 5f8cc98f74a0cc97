#!/bin/bash
# round 5: the blocked path's library tests (fused vs three launches bit for bit; live oracle; C5; float64 path beyond 256 points) + the C5 line
set -o pipefail
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused_block_step or blocked_path or c5_large or float64_path_beyond" --durations=5 > gpurun_out/r05_lgf_tests.log 2>&1 || { tail -30 gpurun_out/r05_lgf_tests.log; exit 1; }
tail -6 gpurun_out/r05_lgf_tests.log
bash tools/r05_c5_prof.sh ${1:-c5_fused2}
