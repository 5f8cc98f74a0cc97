"""The reference ALGORITHM on host cores, for bench.py's ``cpu_baseline`` leg (kind "port").
TEST/BENCH INFRASTRUCTURE ONLY - never imported by the product.

One meta-task exactly as fs_mol/utils/adaptive_dkt_utils.py:361-403 processes it, tasks strictly sequential:
  median-heuristic re-initialisation (adaptive_dkt.py:88-131) -> SciPy L-BFGS-B inner fit of the 3 GP
  hyper-parameters (fit_gpytorch_scipy, adaptive_dkt_utils.py:91) -> dense Hessian + nested-Jacobian mixed
  partials + linalg.solve hypergradient (cauchy_hypergradient.py:43-161, restated in hypergrad_oracle.py)
  with theta = W [d,d] standing in for the feature extractor, float32 like the reference.
The reference's own GPyTorch/BoTorch stack is not installable here or on the GPU box, hence "port".
"""
from __future__ import annotations

import math
import time

import numpy as np
import torch

from . import gp_oracle as O
from .hypergrad_oracle import dense_ift_hypergradient


def one_task(X_s, X_q, y_s, y_q, W, kind: int, dtype=torch.float32, max_inner_evals=None):
    d = W.shape[0]
    X_s, X_q, y_s, y_q = (a.to(dtype) for a in (X_s, X_q, y_s, y_q))
    W = W.to(dtype).clone().requires_grad_(True)
    Zs0 = (X_s @ W / math.sqrt(d)).detach()
    phi0, pri = O.init_phi(Zs0.double(), False, True)
    phi_star, res = O.fit_phi(Zs0, y_s, phi0.to(dtype), pri, kind, dtype=dtype,
                              maxiter=15000 if max_inner_evals is None else max_inner_evals)
    phi = phi_star.to(dtype).clone().requires_grad_(True)

    def f_in(po, pi):
        return O.f_inner(X_s @ po[0] / math.sqrt(d), y_s, pi[0], pri, kind)

    def f_out(po, pi):
        return O.f_outer(X_s @ po[0] / math.sqrt(d), y_s, X_q @ po[0] / math.sqrt(d), y_q, pi[0], kind)

    val = dense_ift_hypergradient(f_out, f_in, (W,), (phi,))
    return val.item(), W.grad, res.nfev


def time_tasks(tasks, kind: int, budget_s: float = 15.0, min_tasks: int = 2, threads=None):
    """Runs tasks sequentially until ``budget_s`` seconds of CPU work are spent; returns (tasks/s, n, cores, nfev)."""
    import os

    # the GPU box gives one GPU a 16-core CPU share; torch with hundreds of threads on 128x128 matrices is
    # pathologically slow (measured: 256 threads -> 100 s/task), so use the cores we actually own
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = threads or min(avail, 16)
    torch.set_num_threads(cores)
    n, nfev = 0, []
    t0 = time.perf_counter()
    while n < tasks.X_s.shape[0] and (n < min_tasks or time.perf_counter() - t0 < budget_s):
        _, _, k = one_task(tasks.X_s[n], tasks.X_q[n], tasks.y_s[n], tasks.y_q[n], tasks.W, kind)
        nfev.append(k)
        n += 1
    dt = time.perf_counter() - t0
    return n / dt, n, cores, float(np.mean(nfev))


# ---- config C3: the same algorithm through the FULL deep-kernel model (GNN + ECFP + fc head) on host cores ---------------------
def one_model_task(model, batch, kind: int):
    """One task of fs_mol/utils/adaptive_dkt_utils.py:361-403 with the default model on the CPU in float32: a forward of the
    extractor on support + query for the re-initialisation, the SciPy L-BFGS-B fit of the three GP parameters on detached
    features (:87-91), then the reference's dense hypergradient with the closures running the WHOLE model, i.e. one more
    forward per Hessian / Jacobian / outer call and h = 3 double-backward passes through the extractor
    (cauchy_hypergradient.py:43-46,78-87,120-121).  ``model`` is this package's torch module on the CPU (pure PyTorch there)."""
    from torch.func import functional_call

    from adkf_ift_amd.models import _Features

    names = [n for n, _ in model.named_parameters() if not n.startswith("gp_")]
    theta = tuple(p.detach().clone().requires_grad_(True) for n, p in model.named_parameters() if not n.startswith("gp_"))
    feat = _Features(model)

    def features(po):
        return functional_call(feat, {"m." + n: p for n, p in zip(names, po)}, (batch,))

    with torch.no_grad():
        Zs0, ys, _, yq = features(theta)
    phi0, pri = O.init_phi(Zs0.double(), bool(model.config.use_numeric_labels), True)
    phi_star, res = O.fit_phi(Zs0, ys, phi0.float(), pri, kind, dtype=torch.float32)
    phi = phi_star.float().clone().requires_grad_(True)

    def f_in(po, pi):
        Zs, y_s, _, _ = features(po)
        return O.f_inner(Zs, y_s, pi[0], pri, kind)

    def f_out(po, pi):
        Zs, y_s, Zq, y_q = features(po)
        return O.f_outer(Zs, y_s, Zq, y_q, pi[0], kind)

    val = dense_ift_hypergradient(f_out, f_in, theta, (phi,))
    return val.item(), res.nfev


def time_model_tasks(model, tasks, kind: int, budget_s: float = 15.0, min_tasks: int = 1, threads=None):
    """C3's ``cpu_baseline``: tasks strictly sequential until ``budget_s`` seconds are spent; returns (tasks/s, n, cores, nfev)."""
    import os

    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = threads or min(avail, 16)
    torch.set_num_threads(cores)
    n, nfev = 0, []
    t0 = time.perf_counter()
    while n < len(tasks) and (n < min_tasks or time.perf_counter() - t0 < budget_s):
        _, k = one_model_task(model, tasks[n], kind)
        nfev.append(k)
        n += 1
    dt = time.perf_counter() - t0
    return n / dt, n, cores, float(np.mean(nfev))
