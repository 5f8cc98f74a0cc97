"""Worker side of tests/test_gpu_stress.py: the float64 oracle quantities of ONE task at the device's fitted point (plus the
same restatement in float32, the conditioning and the oracle's own optimum), computed in CPU-only worker processes so that
the suite's wall time is the device's, not 180 sequential float64 autograd passes (549 s of a 900 s limit in round 2).
Test infrastructure only; never touches the GPU."""
import math
import os

import numpy as np


def worker_init():
    """First thing in every worker, before torch is imported there: hide the GPU, so that the ROCm runtime of a CPU-only worker
    never opens the device (the GPU box allows few processes on its card; these need none of it)."""
    os.environ["HIP_VISIBLE_DEVICES"] = ""
    os.environ["ROCR_VISIBLE_DEVICES"] = ""
    os.environ["CUDA_VISIBLE_DEVICES"] = ""
    import torch
    torch.set_num_threads(2)


def oracle_bundle(args):
    import torch
    from oracle import gp_oracle as O

    zs, ys, zq, yq, phi, kind, regression = args
    torch.set_num_threads(2)
    n, m = zs.shape[0], zq.shape[0]
    p0, opri = O.init_phi(zs.double(), regression, True)
    l0 = float(O.median_lengthscale_init(zs.double()))
    q = O.full_reference_quantities(zs, ys, zq, yq, phi.double(), opri, kind)
    f_star = float(O.f_inner(zs.double(), ys.double(), O.fit_phi(zs.double(), ys.double(), p0, opri, kind)[0], opri, kind))
    noise, os_, ls = O.transform_phi(phi.double())
    A = O.kernel_matrix(zs.double(), zs.double(), os_, ls, kind) + noise * torch.eye(n, dtype=torch.float64)
    cond = max(float(torch.linalg.cond(A)), float(np.linalg.cond(q["pred_cov"])))
    O.DT = torch.float32
    try:
        q32 = O.full_reference_quantities(zs, ys, zq, yq, phi, opri, kind)
    finally:
        O.DT = torch.float64
    keys = ("f_in", "H", "f_out", "g_out", "v", "dZs_total", "dZq_total", "pred_mean", "pred_var")

    def rel(a, ref):
        a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
        return float(np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30))

    e32 = {k: rel(q32[k], q[k]) for k in keys}
    # slack: see the comment in test_gpu_stress.py
    g_floor = 1e-2 * abs(q["f_out"])
    ld_q = float(np.linalg.slogdet(q["pred_cov"])[1])
    quad = 2.0 * q["f_out"] - ld_q - m * math.log(2.0 * math.pi)
    terms = 0.5 * (abs(quad) + abs(ld_q) + m * math.log(2.0 * math.pi))
    slack = {"f_out": max(1.0, terms / abs(q["f_out"])),
             "g_out": max(1.0, g_floor / np.abs(q["g_out"]).max()),
             "v": max(1.0, float(np.abs(np.linalg.inv(q["H"])).sum(1).max()) * max(g_floor, float(np.abs(q["g_out"]).max()))
                      / float(np.abs(q["v"]).max()))}
    return dict(q={k: np.asarray(q[k]) for k in keys}, e32=e32, slack=slack, cond=cond, f_star=f_star, p0=p0.numpy(), l0=l0)
