#!/usr/bin/env python
"""Diagnostic: where one MLL evaluation inside k_inner spends its time.  Builds a SEPARATE library with
-DADKF_EVAL_STAMP=1 (s_memtime at the phase boundaries, workgroup 8, lane 0) and prints the cycle deltas of the last
evaluation of a C2-shaped fit.  s_memtime ticks at 100 MHz on gfx950 (10 ns)."""
import ctypes as C, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
lib_path = "/tmp/libadkf_stamp.so"
LANE = int(sys.argv[1]) if len(sys.argv) > 1 else 0     # which thread of workgroup 8 takes the stamps (0 = wave 0, 448 = wave 7)
subprocess.check_call(["hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-shared", "-fPIC", "-Wno-unused-value", f"-DADKF_EVAL_STAMP={LANE + 1}",
                       "-o", lib_path, os.path.join(ROOT, "adkf_ift_amd/csrc/adkf_gp.hip")])
from adkf_ift_amd import _lib
_lib.LIB_PATH = lib_path
from adkf_ift_amd import gp_ops
from adkf_ift_amd.synthetic import make_tasks
dev = torch.device("cuda:0")
tasks = make_tasks(256, 128, 256)
Zs, Zq = tasks.features()
b = gp_ops.GPBatch(Zs.to(dev), tasks.y_s.to(dev), torch.empty(256, 4, device=dev), "rbf")
phi0, _ = gp_ops.init_params_batch(b)
b.flags = gp_ops.REUSE_DIST
for evals in (20, 21):
    gp_ops.fit(b, phi0, max_evals=evals, exact_evals=True)
    torch.cuda.synchronize()
    lib = _lib.load()
    buf = (C.c_ulonglong * 16)()
    lib.adkf_read_eval_stamps.argtypes = [C.c_void_p]
    assert lib.adkf_read_eval_stamps(buf) == 0
    s = list(buf)
    seq = [(0, 1, "build K"), (1, 10, "barrier"), (10, 2, "sweep"), (2, 3, "solve"), (3, 4, "finish(logdet)"), (4, 5, "traces"),
           (5, 6, "block_sum"), (6, 7, "finalize (wave 0 only)")]
    print(f"evals={evals} lane={LANE} (last SEARCH evaluation; s_memtime ticks; whole evaluation = {s[6] - s[0]} up to block_sum)")
    for i0, i1, n in seq:
        print(f"  {n:24s} {s[i1] - s[i0]:8d}")
