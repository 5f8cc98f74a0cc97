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
