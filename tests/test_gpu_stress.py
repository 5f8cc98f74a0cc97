"""GPU: randomised sweep of the whole path (fresh parameters -> fit to convergence -> IFT hypergradient -> prediction)
over ragged shapes, both kernels and both label types against the float64 oracle evaluated AT THE DEVICE's fitted point,
and long training runs that must stay finite.  Seeds are fixed; every case is reproducible from its printed description."""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-4


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    return torch.device("cuda:0")


def _rel(a, ref):
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30)


def _random_case(rng):
    N = int(rng.choice([4, 6, 9, 16, 23, 32, 48, 64, 100, 128, 129, 160]))
    Nq = int(rng.choice([1, 2, 5, 16, 40, 64, 128, 150, 200]))
    d = int(rng.choice([2, 3, 8, 17, 64, 100, 256, 300]))
    kind = int(rng.integers(0, 2))
    regression = bool(rng.integers(0, 2))
    T = 3
    n_s = [N] + [int(rng.integers(max(3, N // 2), N + 1)) for _ in range(T - 1)]
    n_q = [Nq] + [int(rng.integers(1, Nq + 1)) for _ in range(T - 1)]
    return N, Nq, d, kind, regression, n_s, n_q


@pytest.fixture(scope="module")
def oracle_pool():
    """CPU-only worker processes for the float64 oracle side (fresh interpreters: "spawn"; they never touch the GPU)."""
    import multiprocessing as mp
    from concurrent.futures import ProcessPoolExecutor

    from _stress_oracle import worker_init
    # (four workers: with this process that stays inside the GPU box's bound of 6 processes on the card even if a worker's
    # runtime should open the device in spite of worker_init hiding it)
    pool = ProcessPoolExecutor(max_workers=4, mp_context=mp.get_context("spawn"), initializer=worker_init)
    yield pool
    pool.shutdown(wait=True, cancel_futures=True)


@pytest.mark.parametrize("chunk", range(4))     # 4 x 15 cases, one test per chunk
def test_random_shapes_against_the_oracle(dev, chunk, oracle_pool):
    """The device runs its 15 cases first (fit, hypergradient, prediction); the float64 oracle side of the 45 tasks - evaluated
    AT THE DEVICE's fitted point - is then computed by the worker pool (tests/_stress_oracle.py), which is what used to take
    ~90 s per chunk sequentially."""
    from adkf_ift_amd import gp_ops
    from adkf_ift_amd.synthetic import make_tasks
    from _stress_oracle import oracle_bundle

    rng = np.random.default_rng(20260)
    worst, worst32, failures, jobs = {}, {}, [], []
    n_cmp = n_e32_branch = 0
    for case in range(15 * chunk + 15):
        N, Nq, d, kind, regression, n_s, n_q = _random_case(rng)
        if case < 15 * chunk:
            continue
        desc = dict(case=case, N=N, Nq=Nq, d=d, kind=kind, regression=regression, n_s=n_s, n_q=n_q)
        print("case", desc, flush=True)
        tasks = make_tasks(3, N, d, N_q=Nq, regression=regression, first_task=100 * case)
        Zs, Zq = tasks.features()
        Zs, Zq, ys, yq = Zs.clone(), Zq.clone(), tasks.y_s.clone(), tasks.y_q.clone()
        for t in range(3):   # junk in the padding: must not leak into any result
            Zs[t, n_s[t]:] = 7.5; ys[t, n_s[t]:] = -3.0
            Zq[t, n_q[t]:] = -2.5; yq[t, n_q[t]:] = 9.0
        pri = torch.empty(3, 4, device=dev)
        b = gp_ops.GPBatch(Zs.to(dev), ys.to(dev), pri, kind, Z_q=Zq.to(dev), y_q=yq.to(dev),
                           n_s=torch.tensor(n_s, dtype=torch.int32), n_q=torch.tensor(n_q, dtype=torch.int32))
        phi0, l0 = gp_ops.init_params_batch(b, regression, True)
        b.flags = gp_ops.REUSE_DIST
        phi, f_in, gn, ne, info = gp_ops.fit(b, phi0, 200)
        assert int(info.abs().max()) == 0, (desc, info.tolist())
        assert torch.isfinite(phi).all() and float(phi.abs().max()) < 1e3, (desc, phi.tolist())
        b.flags = gp_ops.REUSE_DIST | gp_ops.REUSE_INNER
        out = gp_ops.ift_hypergrad(b, phi)
        assert int(out["info"].abs().max()) == 0, (desc, out["info"].tolist())
        mean, var, _, info = gp_ops.predict(b, phi)
        assert int(info.abs().max()) == 0
        for t in range(3):
            n, m = n_s[t], n_q[t]
            assert float(out["dZ_s"][t, n:].abs().max() if n < N else 0.0) == 0.0, desc
            assert float(out["dZ_q"][t, m:].abs().max() if m < Nq else 0.0) == 0.0, desc
            got = {"f_in": f_in[t].item(), "H": out["H"][t].cpu().numpy(), "f_out": out["f_out"][t].item(),
                   "g_out": out["g_phi"][t].cpu().numpy(), "v": out["v"][t].cpu().numpy(),
                   "dZs_total": out["dZ_s"][t, :n].cpu().numpy(), "dZq_total": out["dZ_q"][t, :m].cpu().numpy(),
                   "pred_mean": mean[t, :m].cpu().numpy(), "pred_var": var[t, :m].cpu().numpy()}
            fut = oracle_pool.submit(oracle_bundle, (Zs[t, :n].clone(), ys[t, :n].clone(), Zq[t, :m].clone(), yq[t, :m].clone(),
                                                     phi[t].cpu().clone(), kind, regression))
            jobs.append((desc, t, n, m, d, kind, got, l0[t].item(), phi0[t].cpu().numpy(), fut))
    for desc, t, n, m, d, kind, got, l0_dev, phi0_dev, fut in jobs:
        o = fut.result(timeout=600)
        q = o["q"]
        assert abs(l0_dev - o["l0"]) <= 1e-5 * l0_dev, desc
        assert np.abs(phi0_dev - o["p0"]).max() <= 1e-4, desc
        # the optimiser: no worse than the float64 L-BFGS-B optimum (judged on the float64 value AT the device's point)
        assert float(q["f_in"]) <= o["f_star"] + 1e-5 * abs(o["f_star"]) + 2e-6, (desc, t, float(q["f_in"]), got["f_in"], o["f_star"])
        # Tolerance: 1e-4 (north star) on every output of every case - no allowance for ill-conditioning and, since round 3, no
        # "4 x what float32 Cholesky solves reach" branch either: the suite counted how many of its 1 620 comparisons needed that
        # branch - none (worst error / tolerance 0.84) - so it is gone; the float32 restatement's error is still printed.  `slack`
        # only accounts for outputs that are themselves sums of cancelling terms: f_out = (quad + logdet + m log 2pi) / 2 (seen:
        # 0.55 from terms of 58, -114, 57: measured against the size of the terms); grad_phi f_out, a difference of traces of the
        # size of f_out that nearly cancel for a well-fitted task (seen: |g| = 0.046 at f_out = 9.9: held to 1e-4 of
        # max(|g_out|, 0.01 |f_out|)); and v = H^-1 g_out (whatever absolute error g_out is allowed, times |H^-1|_inf).
        well = o["cond"] <= 100.0
        for k, v in got.items():
            e, e32 = _rel(v, q[k]), o["e32"][k]
            tol = TOL * o["slack"].get(k, 1.0)
            n_cmp += 1
            if e > tol and e <= 4.0 * e32:
                n_e32_branch += 1      # (would have passed under round 2's rule: reported, fails all the same)
                print("above the tolerance but within 4 x the float32 restatement's error (case, task, output, err, e32, cond):", desc["case"], t, k, "%.2e" % e, "%.2e" % e32, "%.1e" % o["cond"])
            worst[k] = max(worst.get(k, 0.0), e / tol)
            if e > 0.1 * TOL:
                kk = ("well " if well else "ill ") + k
                worst32[kk] = max(worst32.get(kk, 0.0), float("%.1e" % e))
            if e > tol:
                failures.append((desc["case"], t, k, float("%.2e" % e), float("%.2e" % e32), float("%.1e" % o["cond"]), n, m, d, kind))
    print("worst error / tolerance:", {k: float("%.2f" % v) for k, v in worst.items()})
    print("worst relative error by regime (where > 1e-5):", worst32)
    print("comparisons: %d, all held to 1e-4 x slack (%d of them would have needed round 2's 4 x e32 branch)" % (n_cmp, n_e32_branch))
    for f_ in failures:
        print("FAIL (case, task, output, err, fp32-autograd err, cond, n, m, d, kind):", f_)
    assert not failures, failures[:5]


@pytest.mark.parametrize("kernel, regression, exact", [("rbf", False, True), ("matern", True, True), ("matern", False, False)])
def test_long_training_runs_stay_finite(dev, kernel, regression, exact):
    """300 outer steps at the C1 shape: no task may ever report a failed factorisation or a non-finite number (a fixed
    evaluation budget used to let converged tasks wander off on rounding noise)."""
    from adkf_ift_amd.synthetic import LinearFeatureMap, make_tasks
    from adkf_ift_amd.trainer import ClipAdam, HipGPBackend, MetaStepConfig, meta_step

    tasks = make_tasks(64, 32, 64, regression=regression)
    X_s, X_q, y_s, y_q = (a.to(dev) for a in (tasks.X_s, tasks.X_q, tasks.y_s, tasks.y_q))
    W = tasks.W.to(dev).clone().requires_grad_(True)
    opt = ClipAdam([W], lr=1e-3)
    cfg = MetaStepConfig(gp_kernel=kernel, use_numeric_labels=regression, inner_max_evals=20 if exact else 200,
                         inner_exact_evals=exact, clip_value=1.0)
    feats = LinearFeatureMap(X_s, X_q, W)

    class Spy(HipGPBackend):
        bad = None

        def run(self, *a, **k):
            r = super().run(*a, **k)
            self.flags = torch.stack([(r[4] != 0).any(), (r[5] != 0).any(), ~torch.isfinite(r[0]).all(), ~torch.isfinite(r[1]).all(),
                                      (r[0].abs() > 1e3).any()])
            self.acc = self.flags if getattr(self, "acc", None) is None else (self.acc | self.flags)
            return r

    spy = Spy()
    first = last = None
    for k in range(300):
        losses, _ = meta_step(feats, [W], opt, y_s, y_q, cfg, backend=spy)
        if k == 0:
            first = float(losses.mean())
    last = float(losses.mean())
    assert not bool(spy.acc.any()), spy.acc.tolist()
    assert bool(torch.isfinite(W).all())
    assert last < first, (first, last)     # and the outer objective went down


def test_ill_conditioned_regression_task_is_resolved_stably(dev):
    """16 support / 31 query points in TWO dimensions with regression noise ~0.01: cond(A) = 1.3e3.  With the explicit fp32
    inverse alone this task had f_out off by 2.2e-2 and dL/dZ by 3e-3; the LDL^T re-solve of C and alpha (csrc/ldl.h) brings
    them to ~1e-3 / 1e-4 (f_out = 0.55 is itself a cancellation of terms of size 58, -114 and 57)."""
    from adkf_ift_amd import gp_ops
    from adkf_ift_amd.synthetic import make_tasks
    from oracle import gp_oracle as O

    tasks = make_tasks(3, 16, 2, N_q=64, regression=True, first_task=1800)
    Zs, Zq = tasks.features()
    t, n, m = 1, 15, 31
    zs, ys, zq, yq = Zs[t, :n].contiguous(), tasks.y_s[t, :n].contiguous(), Zq[t, :m].contiguous(), tasks.y_q[t, :m].contiguous()
    b = gp_ops.GPBatch(zs[None].to(dev), ys[None].to(dev), torch.empty(1, 4, device=dev), "rbf", Z_q=zq[None].to(dev), y_q=yq[None].to(dev))
    phi0, _ = gp_ops.init_params_batch(b, True, True)
    phi, _, _, _, info = gp_ops.fit(b, phi0, 200)
    gp_ops.check_info(info)
    out = gp_ops.ift_hypergrad(b, phi)
    gp_ops.check_info(out["info"])
    _, opri = O.init_phi(zs.double(), True, True)
    q = O.full_reference_quantities(zs, ys, zq, yq, phi[0].double().cpu(), opri, 0)
    noise, os_, ls = O.transform_phi(phi[0].double().cpu())
    A = O.kernel_matrix(zs.double(), zs.double(), os_, ls, 0) + noise * torch.eye(n, dtype=torch.float64)
    assert float(torch.linalg.cond(A)) > 500.0          # the regime this test is about
    # f_out = 0.55 is a cancellation of terms of size 58, -114 and 57: the 1e-4 is on the terms
    ld_q = float(np.linalg.slogdet(q["pred_cov"])[1])
    terms = 0.5 * (abs(2.0 * q["f_out"] - ld_q - m * math.log(2.0 * math.pi)) + abs(ld_q) + m * math.log(2.0 * math.pi))
    assert abs(out["f_out"][0].item() - q["f_out"]) <= 1e-4 * terms
    assert _rel(out["dZ_s"][0].cpu().numpy(), q["dZs_total"]) <= 2e-4
    assert _rel(out["dZ_q"][0].cpu().numpy(), q["dZq_total"]) <= 2e-4
    mean, var, _, _ = gp_ops.predict(b, phi)
    assert _rel(mean[0].cpu().numpy(), q["pred_mean"]) <= 1e-4
    assert _rel(var[0].cpu().numpy(), q["pred_var"]) <= 1e-4
    # the same prediction on a batch WITHOUT query labels (models._posterior, evaluate.meta_test and bayes_opt build theirs that
    # way): this task is on the float64 path, whose level 1 - C and mu - must not touch y_q (it read through the null pointer)
    b_nolab = gp_ops.GPBatch(zs[None].to(dev), ys[None].to(dev), b.priors, "rbf", Z_q=zq[None].to(dev), y_q=None)
    mean2, var2, _, info2 = gp_ops.predict(b_nolab, phi)
    torch.cuda.synchronize()
    gp_ops.check_info(info2)
    assert _rel(mean2[0].cpu().numpy(), q["pred_mean"]) <= 1e-4
    assert _rel(var2[0].cpu().numpy(), q["pred_var"]) <= 1e-4


@pytest.mark.parametrize("N, Nq, n_s, n_q", [(300, 320, [300, 262], [320, 290]), (520, 260, [520], [257])])
def test_float64_path_beyond_256_points(dev, N, Nq, n_s, n_q, oracle_pool):
    """Low-dimensional regression tasks (d = 3: ill-conditioned, flagged) with more than 256 support AND query points: the float64
    path's inverses run blocked there (refine64.h r64_inverse_blocked; up to round 2 such tasks stayed on float32 + refinement).
    Every output is held to the same 1e-4 x slack as the random sweep, at the device's fitted point."""
    from adkf_ift_amd import gp_ops
    from adkf_ift_amd.synthetic import make_tasks
    from _stress_oracle import oracle_bundle

    T = len(n_s)
    tasks = make_tasks(T, N, 3, N_q=Nq, regression=True, first_task=4100)
    Zs, Zq = tasks.features()
    Zs, Zq, ys, yq = Zs.clone(), Zq.clone(), tasks.y_s.clone(), tasks.y_q.clone()
    for t in range(T):
        Zs[t, n_s[t]:] = 7.5; ys[t, n_s[t]:] = -3.0
        Zq[t, n_q[t]:] = -2.5; yq[t, n_q[t]:] = 9.0
    b = gp_ops.GPBatch(Zs.to(dev), ys.to(dev), torch.empty(T, 4, device=dev), "matern", Z_q=Zq.to(dev), y_q=yq.to(dev),
                       n_s=torch.tensor(n_s, dtype=torch.int32), n_q=torch.tensor(n_q, dtype=torch.int32))
    phi0, _ = gp_ops.init_params_batch(b, True, True)
    b.flags = gp_ops.REUSE_DIST
    phi, f_in, _, _, info = gp_ops.fit(b, phi0, 200)
    gp_ops.check_info(info)
    b.flags = gp_ops.REUSE_DIST | gp_ops.REUSE_INNER
    out = gp_ops.ift_hypergrad(b, phi)
    gp_ops.check_info(out["info"])
    flagged = gp_ops.double_path_tasks(b).cpu().tolist()
    print("float64 path:", flagged)
    assert sum(flagged) >= 1, "the case is meant to exercise the blocked float64 path"
    mean, var, _, info = gp_ops.predict(b, phi)
    gp_ops.check_info(info)
    futs = [oracle_pool.submit(oracle_bundle, (Zs[t, :n_s[t]].clone(), ys[t, :n_s[t]].clone(), Zq[t, :n_q[t]].clone(), yq[t, :n_q[t]].clone(),
                                               phi[t].cpu().clone(), 1, True)) for t in range(T)]
    for t in range(T):
        n, m = n_s[t], n_q[t]
        o = futs[t].result(timeout=900)
        q = o["q"]
        got = {"f_in": f_in[t].item(), "H": out["H"][t].cpu().numpy(), "f_out": out["f_out"][t].item(), "g_out": out["g_phi"][t].cpu().numpy(),
               "v": out["v"][t].cpu().numpy(), "dZs_total": out["dZ_s"][t, :n].cpu().numpy(), "dZq_total": out["dZ_q"][t, :m].cpu().numpy(),
               "pred_mean": mean[t, :m].cpu().numpy(), "pred_var": var[t, :m].cpu().numpy()}
        errs = {k: (_rel(v, q[k]), TOL * o["slack"].get(k, 1.0)) for k, v in got.items()}
        print("task", t, "flagged", flagged[t], "cond %.1e" % o["cond"], {k: "%.1e/%.1e" % e for k, e in errs.items()})
        assert float(out["dZ_s"][t, n:].abs().max() if n < N else 0.0) == 0.0
        assert float(out["dZ_q"][t, m:].abs().max() if m < Nq else 0.0) == 0.0
        for k, (e, tol) in errs.items():
            assert e <= tol, (t, k, e, tol, o["cond"])
