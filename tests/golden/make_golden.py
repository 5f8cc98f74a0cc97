"""Generates tests/golden/*.npz.  Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Fixtures are DATA: seeded inputs + expected outputs.  Expected GP quantities come from the float64
autograd oracle (oracle/gp_oracle.py); expected theta-gradients of the linear-map cases come from the
REFERENCE's own hypergradient operators, imported by file path from
/root/reference/fs_mol/utils/cauchy_hypergradient.py and cauchy_hypergradient_jvp.py (torch-only
files; the GPyTorch-dependent reference files cannot be imported here - see DESIGN.md).
"""
from __future__ import annotations

import importlib.util
import math
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from oracle import gp_oracle as O  # noqa: E402
from adkf_ift_amd.synthetic import make_tasks  # noqa: E402

REF = "/root/reference/fs_mol/utils"


def _load(name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def gp_case(N, Nq, d, kind, regression, seed, fitted):
    tasks = make_tasks(1, N, d, N_q=Nq, regression=regression, first_task=seed)
    Zs, Zq = tasks.features()
    Zs, Zq, ys, yq = Zs[0], Zq[0], tasks.y_s[0], tasks.y_q[0]
    phi0, pri = O.init_phi(Zs.double(), use_numeric_labels=regression, use_lengthscale_prior=True)
    if fitted:
        phi, res = O.fit_phi(Zs, ys, phi0, pri, kind)
        nit = res.nit
    else:
        # a generic (non-stationary) point: exercises the formulas away from grad f_in = 0
        g = torch.Generator().manual_seed(99 + seed)
        phi = phi0 + 0.3 * torch.randn(3, generator=g, dtype=torch.float64)
        nit = 0
    q = O.full_reference_quantities(Zs, ys, Zq, yq, phi, pri, kind)
    out = dict(Z_s=Zs.numpy(), Z_q=Zq.numpy(), y_s=ys.numpy(), y_q=yq.numpy(), phi=phi.numpy(),
               phi0=phi0.numpy(), priors=pri.as_array(), kind=np.int64(kind), regression=np.int64(regression),
               fitted=np.int64(fitted), fit_nit=np.int64(nit))
    big = N * d >= 8192  # keep the large fixtures small: derivative fields as float32, no duplicates
    for k, v in q.items():
        if k == "pred_cov" or (big and k in ("dZs_direct", "dZq_direct")):
            continue  # direct = total + mixed (support), = total (query)
        v = np.asarray(v)
        out[k] = v.astype(np.float32) if (big and v.size >= 4096) else v
    if N * Nq <= 32 * 32:
        out["pred_cov"] = q["pred_cov"]
    return out


def ard_case(N, Nq, d, kind, regression, seed, fitted):
    """ARD kernel (``use_ard``: one lengthscale per feature dimension, h = 2 + d): same quantities as gp_case, from the
    autograd oracle.  H is kept (h x h) because the HIP path's Hessian-vector products are checked against its columns."""
    tasks = make_tasks(1, N, d, N_q=Nq, regression=regression, first_task=seed)
    Zs, Zq = tasks.features()
    Zs, Zq, ys, yq = Zs[0], Zq[0], tasks.y_s[0], tasks.y_q[0]
    phi0, pri = O.init_phi(Zs.double(), use_numeric_labels=regression, use_lengthscale_prior=True, ard=True)
    if fitted:
        phi, res = O.fit_phi(Zs, ys, phi0, pri, kind)
        nit = res.nit
    else:
        g = torch.Generator().manual_seed(199 + seed)
        phi = phi0 + 0.3 * torch.randn(2 + d, generator=g, dtype=torch.float64)
        nit = 0
    q = O.full_reference_quantities(Zs, ys, Zq, yq, phi, pri, kind)
    out = dict(Z_s=Zs.numpy(), Z_q=Zq.numpy(), y_s=ys.numpy(), y_q=yq.numpy(), phi=phi.numpy(), phi0=phi0.numpy(),
               priors=pri.as_array(), kind=np.int64(kind), regression=np.int64(regression), fitted=np.int64(fitted),
               fit_nit=np.int64(nit), ard=np.int64(1))
    for k, v in q.items():
        if k in ("pred_cov", "dZs_direct", "dZq_direct"):
            continue
        out[k] = np.asarray(v)
    return out


def linear_map_case(N, Nq, d, kind, seed, compact=False):
    """theta = W [d,d]; Z = X W / sqrt(d).  theta.grad from BOTH reference variants.
    compact: the inputs are NOT stored (make_tasks(1, N, d, N_q=Nq, first_task=seed) regenerates them bit for bit on any
    box: CPU generators with fixed seeds) and the gradients are kept as float32 - the C2-shaped case would be 2 MB otherwise."""
    ch = _load("cauchy_hypergradient").cauchy_hypergradient
    chj = _load("cauchy_hypergradient_jvp").cauchy_hypergradient_jvp
    tasks = make_tasks(1, N, d, N_q=Nq, regression=False, first_task=seed)
    Xs, Xq, ys, yq = tasks.X_s[0].double(), tasks.X_q[0].double(), tasks.y_s[0].double(), tasks.y_q[0].double()
    W = tasks.W.double().clone().requires_grad_(True)
    Zs0 = (Xs @ W / math.sqrt(d)).detach()
    phi0, pri = O.init_phi(Zs0, False, True)
    phi_star, _ = O.fit_phi(Zs0, ys, phi0, pri, kind)
    phi = phi_star.clone().requires_grad_(True)

    def f_in(po, pi):
        return O.f_inner(Xs @ po[0] / math.sqrt(d), ys, pi[0], pri, kind)

    def f_out(po, pi):
        return O.f_outer(Xs @ po[0] / math.sqrt(d), ys, Xq @ po[0] / math.sqrt(d), yq, pi[0], kind)

    dev = torch.device("cpu")
    v1 = ch(f_out, f_in, (W,), (phi,), dev)
    g_dense, gphi = W.grad.clone(), phi.grad.clone()
    W.grad = None
    phi.grad = None
    v2 = chj(f_out, f_in, (W,), (phi,), dev)
    g_jvp = W.grad.clone()
    W.grad = None
    phi.grad = None
    v3 = ch(f_out, f_in, (W,), (phi,), dev, ignore_grad_correction=True)
    g_first_order = W.grad.clone()
    assert abs(v1.item() - v2.item()) < 1e-10 and abs(v1.item() - v3.item()) < 1e-10
    if compact:
        assert (g_dense - g_jvp).abs().max().item() <= 1e-6 * g_dense.abs().max().item()
        return dict(N=np.int64(N), Nq=np.int64(Nq), d=np.int64(d), seed=np.int64(seed), phi=phi_star.numpy(),
                    priors=pri.as_array(), kind=np.int64(kind), f_out=np.float64(v1.item()),
                    grad_W_dense=g_dense.numpy().astype(np.float32),   # (the jvp variant agreed to 1e-6 just above)
                    grad_W_first_order=g_first_order.numpy().astype(np.float32), grad_phi=gphi.numpy())
    return dict(X_s=tasks.X_s[0].numpy(), X_q=tasks.X_q[0].numpy(), y_s=tasks.y_s[0].numpy(),
                y_q=tasks.y_q[0].numpy(), W=tasks.W.numpy(), phi=phi_star.numpy(), priors=pri.as_array(),
                kind=np.int64(kind), f_out=np.float64(v1.item()), grad_W_dense=g_dense.numpy(),
                grad_W_jvp=g_jvp.numpy(), grad_W_first_order=g_first_order.numpy(), grad_phi=gphi.numpy())


def harness_case(T, N, d, kind):
    """Fixture for the harness row H (fs_mol/utils/adaptive_dkt_utils.py:352-413): per-task f_out,
    task-mean of hypergradients, and the clip-by-global-norm(1.0) result, with theta = W."""
    ch = _load("cauchy_hypergradient").cauchy_hypergradient
    tasks = make_tasks(T, N, d, regression=False, first_task=500)
    W = tasks.W.double().clone().requires_grad_(True)
    acc = torch.zeros_like(W)
    f_outs, phis, pris = [], [], []
    for t in range(T):
        Xs, Xq, ys, yq = (a[t].double() for a in (tasks.X_s, tasks.X_q, tasks.y_s, tasks.y_q))
        Zs0 = (Xs @ W / math.sqrt(d)).detach()
        phi0, pri = O.init_phi(Zs0, False, True)
        phi_star, _ = O.fit_phi(Zs0, ys, phi0, pri, kind)
        phi = phi_star.clone().requires_grad_(True)

        def f_in(po, pi):
            return O.f_inner(Xs @ po[0] / math.sqrt(d), ys, pi[0], pri, kind)

        def f_out(po, pi):
            return O.f_outer(Xs @ po[0] / math.sqrt(d), ys, Xq @ po[0] / math.sqrt(d), yq, pi[0], kind)

        val = ch(f_out, f_in, (W,), (phi,), torch.device("cpu"))
        acc += W.grad.clone() / T
        f_outs.append(val.item())
        phis.append(phi_star.numpy())
        pris.append(pri.as_array())
    norm = acc.norm().item()
    clipped = acc * min(1.0, 1.0 / (norm + 1e-6))
    if d >= 128:   # C2-shaped: float32 gradients keep the fixture under 1 MB
        acc, clipped = acc.float(), clipped.float()
    return dict(T=np.int64(T), N=np.int64(N), d=np.int64(d), kind=np.int64(kind), phi=np.stack(phis),
                priors=np.stack(pris), f_out=np.array(f_outs), grad_mean=acc.numpy(), grad_norm=np.float64(norm),
                grad_clipped=clipped.numpy())


def noise_floor_case():
    """One support set on which the fp32 device optimiser, run for a FIXED number of evaluations, used to iterate on
    past convergence and leave to outputscale 1e8.  Z_s / y_s were captured on the GPU (tools/capture_fit_divergence.py:
    task 142 of the C2 benchmark after 88 outer steps) and are kept as they are; phi_star is the oracle's fit."""
    old = np.load(os.path.join(HERE, "fit_noise_floor_task.npz"))
    Zs, ys = torch.tensor(old["Z_s"]).double(), torch.tensor(old["y_s"]).double()
    phi0, pri = O.init_phi(Zs, False, True)
    phi = O.fit_phi(Zs, ys, phi0, pri, 0)[0]
    return dict(Z_s=old["Z_s"], y_s=old["y_s"], phi_star=phi.numpy())


def main():
    torch.manual_seed(0)
    cases = []
    for (N, Nq, d) in [(8, 8, 4), (16, 32, 16), (32, 32, 64), (128, 128, 256)]:
        for kind in (O.KERNEL_RBF, O.KERNEL_MATERN52):
            for regression in (0, 1):
                seeds = (0, 1) if N < 128 else (0,)
                for seed in seeds:
                    fitted = 1 if seed == 0 else 0
                    name = f"gp_N{N}_Nq{Nq}_d{d}_k{kind}_r{regression}_s{seed}"
                    cases.append((name, lambda a=(N, Nq, d, kind, regression, seed, fitted): gp_case(*a)))
    # ragged / small edge cases
    cases.append(("gp_N5_Nq3_d7_k0_r0_s2", lambda: gp_case(5, 3, 7, 0, 0, 2, 0)))
    cases.append(("gp_N17_Nq41_d33_k1_r1_s3", lambda: gp_case(17, 41, 33, 1, 1, 3, 0)))
    cases.append(("gp_N64_Nq128_d96_k0_r0_s4", lambda: gp_case(64, 128, 96, 0, 0, 4, 1)))
    for kind in (0, 1):
        cases.append((f"linmap_N16_Nq24_d12_k{kind}", lambda k=kind: linear_map_case(16, 24, 12, k, 7)))
    cases.append(("harness_T4_N16_d8_k0", lambda: harness_case(4, 16, 8, 0)))
    # the headline shape (C2: N = N_q = 128, d = 256) through the reference's operators
    cases.append(("linmap_N128_Nq128_d256_k0", lambda: linear_map_case(128, 128, 256, 0, 7, compact=True)))
    cases.append(("linmap_N128_Nq128_d256_k1", lambda: linear_map_case(128, 128, 256, 1, 8, compact=True)))
    cases.append(("harness_T4_N128_d256_k0", lambda: harness_case(4, 128, 256, 0)))
    for a in [(8, 8, 4, 0, 0, 0, 1), (16, 24, 12, 1, 0, 1, 0), (32, 32, 16, 0, 1, 0, 1), (48, 40, 24, 1, 0, 2, 1),
              (128, 128, 64, 1, 0, 0, 0)]:
        cases.append(("ard_N%d_Nq%d_d%d_k%d_r%d_s%d" % a[:6], lambda a=a: ard_case(*a)))
    cases.append(("fit_noise_floor_task", noise_floor_case))
    only = sys.argv[1] if len(sys.argv) > 1 else ""     # python make_golden.py [name-prefix]
    for name, fn in cases:
        if not name.startswith(only):
            continue
        out = fn()
        np.savez(os.path.join(HERE, name + ".npz"), **out)
        print(name, "ok", flush=True)


if __name__ == "__main__":
    main()
