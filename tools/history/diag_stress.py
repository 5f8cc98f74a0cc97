#!/usr/bin/env python
"""Diagnostic (GPU): where do the stress-test errors of selected cases come from?  For each case the pipeline runs
(a) as shipped, (b) with the squared distances in the workspace REPLACED by float64-exact ones rounded to float32
(written straight into the caller-owned workspace, then ADKF_BATCH_REUSE_DIST), and prints the errors against the float64
oracle at the device's fitted point.  Usage: python tools/diag_stress.py 7 42 44 51"""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from test_gpu_stress import _random_case, _rel  # noqa: E402
from adkf_ift_amd import gp_ops  # noqa: E402
from adkf_ift_amd.synthetic import make_tasks  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402


def au(nfloat):
    return (nfloat * 4 + 255) & ~255


def d2_offsets(T, ns, nq, d):
    off = 0
    off += au(T * d)            # mean
    o_ss = off
    off += au(T * ns * ns)
    o_qs = off
    off += au(T * nq * ns)
    o_qq = off
    return o_ss, o_qs, o_qq


def main():
    want = [int(a) for a in sys.argv[1:]] or [7, 42, 44, 51]
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(20260)
    for case in range(max(want) + 1):
        N, Nq, d, kind, regression, n_s, n_q = _random_case(rng)
        if case not in want:
            continue
        tasks = make_tasks(3, N, d, N_q=Nq, regression=regression, first_task=100 * case)
        Zs, Zq = tasks.features()
        Zs, Zq, ys, yq = Zs.clone(), Zq.clone(), tasks.y_s.clone(), tasks.y_q.clone()
        for t in range(3):
            Zs[t, n_s[t]:] = 7.5; ys[t, n_s[t]:] = -3.0
            Zq[t, n_q[t]:] = -2.5; yq[t, n_q[t]:] = 9.0
        for mode in ("shipped", "exact-D2"):
            pri = torch.empty(3, 4, device=dev)
            b = gp_ops.GPBatch(Zs.to(dev), ys.to(dev), pri, kind, Z_q=Zq.to(dev), y_q=yq.to(dev),
                               n_s=torch.tensor(n_s, dtype=torch.int32), n_q=torch.tensor(n_q, dtype=torch.int32))
            phi0, l0 = gp_ops.init_params_batch(b, regression, True)
            if mode == "exact-D2":
                ws, _ = b.workspace()
                o_ss, o_qs, o_qq = d2_offsets(3, N, Nq, d)
                zs64, zq64 = Zs.double(), Zq.double()
                D_ss = (torch.cdist(zs64, zs64) ** 2).float()
                D_qs = (torch.cdist(zq64, zs64) ** 2).float()
                D_qq = (torch.cdist(zq64, zq64) ** 2).float()
                for t in range(3):
                    D_ss[t].fill_diagonal_(0.0)
                    D_qq[t].fill_diagonal_(0.0)
                for off, D in ((o_ss, D_ss), (o_qs, D_qs), (o_qq, D_qq)):
                    ws[off:off + D.numel() * 4].view(torch.float32).copy_(D.reshape(-1).to(dev))
            b.flags = gp_ops.REUSE_DIST
            phi, f_in, gn, ne, info = gp_ops.fit(b, phi0, 200)
            b.flags = gp_ops.REUSE_DIST | gp_ops.REUSE_INNER
            out = gp_ops.ift_hypergrad(b, phi)
            mean, var, _, _ = gp_ops.predict(b, phi)
            for t in range(3):
                n, m = n_s[t], n_q[t]
                zs, zq = Zs[t, :n], Zq[t, :m]
                _, opri = O.init_phi(zs.double(), regression, True)
                q = O.full_reference_quantities(zs, ys[t, :n], zq, yq[t, :m], phi[t].double().cpu(), opri, kind)
                got = {"f_in": f_in[t].item(), "H": out["H"][t].cpu().numpy(), "f_out": out["f_out"][t].item(),
                       "g_out": out["g_phi"][t].cpu().numpy(), "v": out["v"][t].cpu().numpy(),
                       "dZs": out["dZ_s"][t, :n].cpu().numpy(), "dZq": out["dZ_q"][t, :m].cpu().numpy(),
                       "mean": mean[t, :m].cpu().numpy(), "var": var[t, :m].cpu().numpy()}
                ref = {"f_in": q["f_in"], "H": q["H"], "f_out": q["f_out"], "g_out": q["g_out"], "v": q["v"], "dZs": q["dZs_total"],
                       "dZq": q["dZq_total"], "mean": q["pred_mean"], "var": q["pred_var"]}
                print("case %d task %d (n=%d m=%d d=%d kind=%d) %-9s" % (case, t, n, m, d, kind, mode),
                      " ".join("%s %.1e" % (k, _rel(got[k], ref[k])) for k in got), flush=True)


if __name__ == "__main__":
    main()
