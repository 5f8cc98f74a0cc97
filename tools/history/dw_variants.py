#!/usr/bin/env python
"""The backward product of the synthetic feature map, dW = X^T dZ (256 x 256 output, 65536 rows reduced), as a chunked bmm +
sum for several chunk counts and as one GEMM, each with the library algorithm measured by TunableOp (run through gpurun):
which form the bench's harness should use.  Not part of the library."""
import os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adkf_ift_amd.gemm_tuning import use_tuned_gemms

dev = torch.device("cuda:0")
use_tuned_gemms(os.path.join(tempfile.mkdtemp(), "t.csv"), tune=True)
R, d = 2 * 256 * 128, 256
X = torch.randn(R, d, device=dev)
g = torch.randn(R, d, device=dev)

def timed(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3

print("one GEMM X^T g: %.1f us" % timed(lambda: X.t() @ g))
for ch in (2, 4, 8, 16, 32, 64, 128):
    Xv, gv = X.view(ch, R // ch, d).transpose(1, 2), g.view(ch, R // ch, d)
    print("chunks %3d: bmm %.1f us, bmm + sum %.1f us" % (ch, timed(lambda: torch.bmm(Xv, gv)), timed(lambda: torch.bmm(Xv, gv).sum(0))))
print("forward X W: %.1f us" % timed(lambda: X @ g[:d]))
