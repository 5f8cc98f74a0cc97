"""GPU parity of the HIP path against the committed golden vectors (float64 autograd oracle, with the
linear-map theta-gradients produced by the REFERENCE's cauchy_hypergradient files).  All calls go through the
C ABI (ctypes).  Tolerance: 1e-4 relative (BASELINE.json north_star) on log-marginal-likelihood and IFT
gradients; matrices/vectors are compared in max-norm relative to the largest reference entry."""
import glob
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-4


def rel(a, ref):
    a = np.asarray(a, dtype=np.float64)
    ref = np.asarray(ref, dtype=np.float64)
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30)


def _cases(golden_dir):
    return sorted(glob.glob(os.path.join(golden_dir, "gp_*.npz")))


def _batch(g, dev, pad_s=0, pad_q=0):
    from adkf_ift_amd import gp_ops

    def pad(x, p):
        x = torch.as_tensor(x, dtype=torch.float32)
        if p:
            x = torch.cat([x, torch.full((p, *x.shape[1:]), 7.5)])  # padding rows hold junk on purpose
        return x[None].to(dev)

    n, m = g["Z_s"].shape[0], g["Z_q"].shape[0]
    b = gp_ops.GPBatch(pad(g["Z_s"], pad_s), pad(g["y_s"], pad_s), torch.tensor(g["priors"], dtype=torch.float32)[None].to(dev),
                       int(g["kind"]), Z_q=pad(g["Z_q"], pad_q), y_q=pad(g["y_q"], pad_q),
                       n_s=torch.tensor([n]) if pad_s else None, n_q=torch.tensor([m]) if pad_q else None)
    phi = torch.tensor(g["phi"], dtype=torch.float32)[None].to(dev)
    return b, phi, n, m


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    return torch.device("cuda:0")


def test_all_golden_cases(golden_dir, dev):
    from adkf_ift_amd import gp_ops

    files = _cases(golden_dir)
    assert len(files) >= 30
    worst = {}
    for f in files:
        g = np.load(f)
        for pad_s, pad_q in ((0, 0), (3, 5)):   # (the padded 128-point cases, 131 x 133, take the blocked path of large.h)
            b, phi, n, m = _batch(g, dev, pad_s, pad_q)
            l0 = gp_ops.median_lengthscale(b)
            fin, gin, dZin, info = gp_ops.mll_value_grad(b, phi, want_dZ=True)
            gp_ops.check_info(info)
            out = gp_ops.ift_hypergrad(b, phi)
            gp_ops.check_info(out["info"])
            mean, var, cov, info = gp_ops.predict(b, phi, want_cov=True)
            got = {
                "l0": l0[0].item(), "f_in": fin[0].item(), "g_in": gin[0].cpu().numpy(), "dfin_dZs": dZin[0, :n].cpu().numpy(),
                "H": out["H"][0].cpu().numpy(), "f_out": out["f_out"][0].item(), "g_out": out["g_phi"][0].cpu().numpy(),
                "v": out["v"][0].cpu().numpy(), "dZs_total": out["dZ_s"][0, :n].cpu().numpy(),
                "dZq_total": out["dZ_q"][0, :m].cpu().numpy(), "pred_mean": mean[0, :m].cpu().numpy(),
                "pred_var": var[0, :m].cpu().numpy(),
            }
            if "pred_cov" in g.files:
                got["pred_cov"] = cov[0, :m, :m].cpu().numpy()
            for k, v in got.items():
                ref = g[k]
                e = rel(v, ref)
                if k == "g_in" and int(g["fitted"]):
                    # at a fitted point grad f_in ~ 0: compare against the gradient scale of the start point instead
                    e = np.abs(np.asarray(v) - ref).max() / 1e-1
                worst[k] = max(worst.get(k, 0.0), e)
                assert e <= TOL, (os.path.basename(f), pad_s, k, e)
            if pad_s:
                assert float(out["dZ_s"][0, n:].abs().max()) == 0.0 and float(out["dZ_q"][0, m:].abs().max()) == 0.0
    print("worst relative errors:", {k: float("%.2e" % v) for k, v in worst.items()})


def test_flags_and_outer_only(golden_dir, dev):
    """ignore_grad_correction -> first-order gradient; ignore_direct_grad -> only -mixed (cauchy_hypergradient.py:11-13)."""
    from adkf_ift_amd import gp_ops

    g = np.load(os.path.join(golden_dir, "gp_N32_Nq32_d64_k1_r0_s1.npz"))
    b, phi, n, m = _batch(g, dev)
    o1 = gp_ops.ift_hypergrad(b, phi, ignore_grad_correction=True)
    assert rel(o1["dZ_s"][0].cpu().numpy(), g["dZs_direct"]) <= TOL
    assert rel(o1["dZ_q"][0].cpu().numpy(), g["dZq_direct"]) <= TOL
    o2 = gp_ops.ift_hypergrad(b, phi, ignore_direct_grad=True)
    assert rel(o2["dZ_s"][0].cpu().numpy(), -g["mixed_Zs"]) <= TOL
    assert float(o2["dZ_q"].abs().max()) == 0.0
    f, gphi, dZs, dZq, info = gp_ops.outer_nll_value_grad(b, phi)
    assert rel(f[0].item(), g["f_out"]) <= TOL and rel(gphi[0].cpu().numpy(), g["g_out"]) <= TOL
    assert rel(dZs[0].cpu().numpy(), g["dZs_direct"]) <= TOL


def test_batched_equals_single(golden_dir, dev):
    """A meta-batch of different tasks gives, per task, exactly what the task gives alone (bitwise)."""
    from adkf_ift_amd import gp_ops
    from adkf_ift_amd.synthetic import make_tasks

    tasks = make_tasks(11, 32, 24, N_q=40)
    Zs, Zq = tasks.features()
    Zs, Zq, ys, yq = Zs.to(dev), Zq.to(dev), tasks.y_s.to(dev), tasks.y_q.to(dev)
    phi, pri, _ = gp_ops.init_params(Zs)
    b = gp_ops.GPBatch(Zs, ys, pri, "matern", Z_q=Zq, y_q=yq)
    full = gp_ops.ift_hypergrad(b, phi)
    for t in (0, 5, 10):
        bt = gp_ops.GPBatch(Zs[t:t + 1], ys[t:t + 1], pri[t:t + 1], "matern", Z_q=Zq[t:t + 1], y_q=yq[t:t + 1])
        one = gp_ops.ift_hypergrad(bt, phi[t:t + 1])
        for k in ("f_out", "dZ_s", "dZ_q", "v"):
            assert torch.equal(one[k][0], full[k][t]), (t, k)


def test_fit_reaches_oracle_optimum(golden_dir, dev):
    """Inner optimiser judged separately: f_in(phi*) and max|grad| vs the float64 L-BFGS-B oracle."""
    from adkf_ift_amd import gp_ops

    for name in ("gp_N32_Nq32_d64_k0_r0_s0", "gp_N32_Nq32_d64_k1_r1_s0", "gp_N128_Nq128_d256_k0_r0_s0",
                 "gp_N128_Nq128_d256_k1_r1_s0", "gp_N64_Nq128_d96_k0_r0_s4", "gp_N8_Nq8_d4_k0_r0_s0"):
        g = np.load(os.path.join(golden_dir, name + ".npz"))
        assert int(g["fitted"])
        b, _, n, m = _batch(g, dev)
        phi0 = torch.tensor(g["phi0"], dtype=torch.float32)[None].to(dev)
        phi, f, gn, ne, info = gp_ops.fit(b, phi0, max_evals=200)
        gp_ops.check_info(info)
        assert f[0].item() <= g["f_in"] + 1e-5 * abs(g["f_in"]) + 1e-6, (name, f[0].item(), float(g["f_in"]))
        # The fit stops when f no longer decreases in float32.  With curvature h <= 0.4 (these fixtures) a gradient below
        # sqrt(2 h eps32 f) ~ 3.5e-4 changes f by less than one float32 ulp, so that is what "converged" can resolve; which
        # side of 2e-4 a run ends on depends on the rounding of the reductions (seen: 4e-7 and 2.4e-4 for the same f).
        assert gn[0].item() <= 5e-4, (name, gn[0].item())
        assert ne[0].item() <= 200


def test_init_params_matches_reference_recipe(golden_dir, dev):
    from adkf_ift_amd import gp_ops

    g = np.load(os.path.join(golden_dir, "gp_N16_Nq32_d16_k0_r1_s0.npz"))
    Zs = torch.tensor(g["Z_s"])[None].to(dev)
    phi, pri, l0 = gp_ops.init_params(Zs, use_numeric_labels=True, use_lengthscale_prior=True)
    assert rel(phi[0].cpu().numpy(), g["phi0"]) <= 1e-5
    assert rel(pri[0].cpu().numpy(), g["priors"]) <= 1e-5


def test_not_positive_definite_is_reported(dev):
    """Duplicate points with (almost) no noise: the factorisation must flag the task, not add jitter."""
    from adkf_ift_amd import gp_ops

    Z = torch.zeros(1, 8, 4, device=dev)
    y = torch.ones(1, 8, device=dev)
    pri = torch.tensor([[-2.0, 0.25, 0.0, -1.0]], device=dev)
    b = gp_ops.GPBatch(Z, y, pri, "rbf")
    phi = torch.tensor([[-1e4, 0.0, 0.0]], device=dev)  # softplus(-1e4) = 0 -> noise = 1e-4, K = ln2 * ones
    f, g, _, info = gp_ops.mll_value_grad(b, phi)
    # rank-one K + 1e-4 I is still PD in exact arithmetic; make it singular by exact cancellation instead
    Zbad = torch.zeros(1, 4, 2, device=dev)
    bbad = gp_ops.GPBatch(Zbad, torch.ones(1, 4, device=dev), pri, "rbf")
    phibad = torch.tensor([[-1e4, 80.0, 0.0]], device=dev)  # outputscale 80 >> noise 1e-4: fp32 Schur pivots hit <= 0
    f, g, _, info = gp_ops.mll_value_grad(bbad, phibad)
    if int(info[0]) != 0:
        with pytest.raises(RuntimeError):
            gp_ops.check_info(info)


def test_argument_errors(dev):
    from adkf_ift_amd import gp_ops

    with pytest.raises(RuntimeError):
        gp_ops.GPBatch(torch.zeros(1, 4, 2), torch.zeros(1, 4), torch.zeros(1, 4))  # CPU tensors: no fallback
    with pytest.raises(ValueError):
        gp_ops.kernel_id("cossim")
    big = gp_ops.GPBatch(torch.zeros(1, 4100, 2, device=dev), torch.zeros(1, 4100, device=dev), torch.zeros(1, 4, device=dev))
    with pytest.raises(RuntimeError):
        gp_ops.median_lengthscale(big)


@pytest.mark.parametrize("N,Nq,d,kernel", [(200, 256, 64, "rbf"), (256, 160, 48, "matern"), (130, 129, 32, "rbf"),
                                           (16, 256, 64, "matern"), (384, 300, 64, "rbf"), (516, 132, 40, "matern")])
def test_blocked_path_against_live_oracle(dev, N, Nq, d, kernel):
    """More than 128 support or query points: the blocked sweep (csrc/large.h).  No committed fixture, so the float64
    oracle is evaluated live (tests may call the oracle) on seeded synthetic tasks with ragged sizes - including tasks
    whose real size would fit the register path and sizes that are not multiples of the 128-pivot block."""
    from adkf_ift_amd import gp_ops
    from adkf_ift_amd.synthetic import make_tasks
    from oracle import gp_oracle as O

    T = 3
    tasks = make_tasks(T, N, d, N_q=Nq, first_task=40)
    Zs, Zq = tasks.features()
    n_s = torch.tensor([N, N - 7, max(N - 64, 5)])
    n_q = torch.tensor([Nq, Nq - 1, Nq - 100])
    kind = gp_ops.kernel_id(kernel)
    pri = torch.empty(T, 4, device=dev)
    b = gp_ops.GPBatch(Zs.to(dev), tasks.y_s.to(dev), pri, kernel, Z_q=Zq.to(dev), y_q=tasks.y_q.to(dev), n_s=n_s, n_q=n_q)
    phi0, l0 = gp_ops.init_params_batch(b)
    b.flags = gp_ops.REUSE_DIST
    phi, f, gn, ne, info = gp_ops.fit(b, phi0, max_evals=60)
    gp_ops.check_info(info)
    b.flags = gp_ops.REUSE_DIST | gp_ops.REUSE_INNER
    out = gp_ops.ift_hypergrad(b, phi)
    gp_ops.check_info(out["info"])
    for t in range(T):
        ns, nq = int(n_s[t]), int(n_q[t])
        p = O.Priors(*pri[t].double().cpu().tolist())
        l0_ref = O.median_lengthscale_init(Zs[t, :ns].double()).item()
        assert abs(l0[t].item() - l0_ref) <= 1e-5 * l0_ref
        q = O.full_reference_quantities(Zs[t, :ns], tasks.y_s[t, :ns], Zq[t, :nq], tasks.y_q[t, :nq], phi[t].double().cpu(), p, kind)
        assert rel(f[t].item(), q["f_in"]) <= TOL and rel(out["f_out"][t].item(), q["f_out"]) <= TOL
        assert rel(out["H"][t].cpu().numpy(), q["H"]) <= TOL
        assert rel(out["dZ_s"][t, :ns].cpu().numpy(), q["dZs_total"]) <= TOL
        assert rel(out["dZ_q"][t, :nq].cpu().numpy(), q["dZq_total"]) <= TOL
        assert float(out["dZ_s"][t, ns:].abs().max() if ns < N else 0.0) == 0.0
        assert gn[t].item() <= 5e-4


@pytest.mark.parametrize("N,Nq,d,kernel", [(516, 132, 40, "matern"), (384, 300, 64, "rbf"), (1024, 1024, 64, "rbf"), (250, 130, 24, "rbf")])
def test_fused_block_step_equals_three_launches(dev, N, Nq, d, kernel):
    """csrc/large_fused.h (the update of block step k and the diagonal sweep of block step k + 1 in one launch, the next diagonal
    block handed from the tile workgroups to the sweeping one through write-through stores, an agent-scope arrival counter and one
    acquire; batch flag ADKF_BATCH_LG_FUSED) against the three launches per block step of csrc/large.h (ADKF_BATCH_LG_UNFUSED): the same arithmetic in
    the same order, so EVERY output of the fit and of the hypergradient stage must be equal bit for bit - a stale read of the
    handed-over block would show up as a difference.  Ragged sizes, block counts from 2 to 8, sizes that are not a multiple of the
    64-point tiles; run three times (a visibility bug comes and goes)."""
    from adkf_ift_amd import gp_ops
    from adkf_ift_amd.synthetic import make_tasks

    T = 5
    tasks = make_tasks(T, N, d, N_q=Nq, first_task=140)
    Zs, Zq = tasks.features()
    n_s = torch.tensor([N, N - 7, max(N - 64, 5), max(N - 129, 3), N - 1])
    n_q = torch.tensor([Nq, Nq - 1, max(Nq - 100, 2), Nq, max(Nq - 65, 1)])

    def run(flag):
        pri = torch.empty(T, 4, device=dev)
        b = gp_ops.GPBatch(Zs.to(dev), tasks.y_s.to(dev), pri, kernel, Z_q=Zq.to(dev), y_q=tasks.y_q.to(dev), n_s=n_s, n_q=n_q)
        b.flags = flag
        phi0, _ = gp_ops.init_params_batch(b)
        b.flags = gp_ops.REUSE_DIST | flag
        phi, f, gn, ne, info = gp_ops.fit(b, phi0, max_evals=12, exact_evals=True)
        gp_ops.check_info(info)
        b.flags = gp_ops.REUSE_DIST | gp_ops.REUSE_INNER | flag
        out = gp_ops.ift_hypergrad(b, phi)
        gp_ops.check_info(out["info"])
        return [phi, f, gn, out["f_out"], out["H"], out["v"], out["dZ_s"], out["dZ_q"]]

    want = run(gp_ops.LG_UNFUSED)
    for rep in range(3):
        got = run(gp_ops.LG_FUSED)
        for k, (a, b_) in enumerate(zip(got, want)):
            assert torch.equal(a, b_), (rep, k, (a - b_).abs().max().item())


def test_c5_large_support_regime(dev):
    """BASELINE.json config C5: N_support = 1024, d = 512 (one task against the live float64 oracle)."""
    from adkf_ift_amd import gp_ops
    from adkf_ift_amd.synthetic import make_tasks
    from oracle import gp_oracle as O

    N, Nq, d, T = 1024, 1024, 512, 2
    tasks = make_tasks(T, N, d, N_q=Nq, first_task=7)
    Zs, Zq = tasks.features()
    pri = torch.empty(T, 4, device=dev)
    b = gp_ops.GPBatch(Zs.to(dev), tasks.y_s.to(dev), pri, "rbf", Z_q=Zq.to(dev), y_q=tasks.y_q.to(dev))
    phi0, l0 = gp_ops.init_params_batch(b)
    b.flags = gp_ops.REUSE_DIST
    phi, f, gn, ne, info = gp_ops.fit(b, phi0, max_evals=40)
    gp_ops.check_info(info)
    b.flags = gp_ops.REUSE_DIST | gp_ops.REUSE_INNER
    out = gp_ops.ift_hypergrad(b, phi)
    gp_ops.check_info(out["info"])
    t = 1
    p = O.Priors(*pri[t].double().cpu().tolist())
    l0_ref = O.median_lengthscale_init(Zs[t].double()).item()
    assert abs(l0[t].item() - l0_ref) <= 1e-5 * l0_ref
    q = O.full_reference_quantities(Zs[t], tasks.y_s[t], Zq[t], tasks.y_q[t], phi[t].double().cpu(), p, 0)
    assert rel(f[t].item(), q["f_in"]) <= TOL and rel(out["f_out"][t].item(), q["f_out"]) <= TOL
    assert rel(out["H"][t].cpu().numpy(), q["H"]) <= TOL
    assert rel(out["v"][t].cpu().numpy(), q["v"]) <= TOL
    assert rel(out["dZ_s"][t].cpu().numpy(), q["dZs_total"]) <= TOL
    assert rel(out["dZ_q"][t].cpu().numpy(), q["dZq_total"]) <= TOL


def test_fixed_evaluation_budget_does_not_iterate_past_convergence(golden_dir, dev):
    """exact_evals mode (the benchmark's "exactly I evaluations") must stop MOVING once converged and spend the rest of
    its budget at the optimum: quasi-Newton updates built from differences at the fp32 noise floor once sent the
    captured task to outputscale 1e8 at evaluation 20, where the fp32 value is garbage that passes the Armijo test."""
    from adkf_ift_amd import gp_ops
    from adkf_ift_amd.synthetic import make_tasks

    g = np.load(os.path.join(golden_dir, "fit_noise_floor_task.npz"))
    Zs = torch.tensor(g["Z_s"])[None].to(dev)
    ys = torch.tensor(g["y_s"])[None].to(dev)
    pri = torch.empty(1, 4, device=dev)
    b = gp_ops.GPBatch(Zs, ys, pri, "rbf")
    phi0, _ = gp_ops.init_params_batch(b, False, True)
    ref, f_ref, _, ne_ref, info = gp_ops.fit(b, phi0, 200)
    gp_ops.check_info(info)
    assert np.abs(ref[0].cpu().numpy() - g["phi_star"]).max() <= 2e-3          # the float64 L-BFGS-B optimum
    for budget in (14, 20, 21, 33, 60):
        phi, f, gn, ne, info = gp_ops.fit(b, phi0, budget, exact_evals=True)
        gp_ops.check_info(info)
        assert int(ne[0]) == budget
        assert torch.isfinite(phi).all() and (phi - ref).abs().max().item() <= 1e-3, (budget, phi.tolist())
        assert abs(f[0].item() - f_ref[0].item()) <= 1e-6

    # the same property over a whole meta-batch: any budget >= the evaluations a task needs returns its optimum
    tasks = make_tasks(64, 64, 32, N_q=8)
    Zs = (tasks.X_s @ tasks.W / math.sqrt(32)).to(dev).contiguous()
    b = gp_ops.GPBatch(Zs, tasks.y_s.to(dev), torch.empty(64, 4, device=dev), "matern")
    phi0, _ = gp_ops.init_params_batch(b, False, True)
    ref, f_ref, _, ne_ref, info = gp_ops.fit(b, phi0, 200)
    gp_ops.check_info(info)
    for budget in (40, 80):
        phi, f, gn, ne, info = gp_ops.fit(b, phi0, budget, exact_evals=True)
        gp_ops.check_info(info)
        done = ne_ref <= budget
        assert done.any() and (ne == budget).all()
        assert torch.isfinite(phi).all()
        assert (phi[done] - ref[done]).abs().max().item() <= 1e-3
        assert (f[done] - f_ref[done]).abs().max().item() <= 1e-6


@pytest.mark.parametrize("kernel", ["rbf", "matern"])
def test_two_tasks_per_cu_fit_equals_the_resident_one(dev, kernel):
    """More tasks than the chip has CUs: adkf_fit takes the <= 128-register build of k_inner (two workgroups per CU, D^2 in LDS,
    kappa'(u) u formed again in the trace pass - csrc/inner.h).  Same arithmetic in the same order, so the same tasks fitted in
    chunks that stay below the CU count (the resident build) must give the same numbers; ragged sizes included."""
    from adkf_ift_amd import gp_ops
    from adkf_ift_amd.synthetic import make_tasks

    T, N, d = 2 * torch.cuda.get_device_properties(dev).multi_processor_count // 2 + 44, 128, 32      # 300 on an MI355X
    tasks = make_tasks(T, N, d, regression=(kernel == "matern"), first_task=7000)
    Zs, _ = tasks.features()
    n_s = torch.full((T,), N, dtype=torch.int32)
    n_s[::7] = 97
    n_s[3::11] = 66
    Zs, ys = Zs.to(dev), tasks.y_s.to(dev)

    def run(lo, hi):
        pri = torch.empty(hi - lo, 4, device=dev)
        b = gp_ops.GPBatch(Zs[lo:hi].contiguous(), ys[lo:hi].contiguous(), pri, kernel, n_s=n_s[lo:hi].contiguous())
        phi0, _ = gp_ops.init_params_batch(b, kernel == "matern", True)
        b.flags = gp_ops.REUSE_DIST
        phi, f, gn, ne, info = gp_ops.fit(b, phi0, 60)
        gp_ops.check_info(info)
        return phi, f, ne

    phi_all, f_all, ne_all = run(0, T)
    parts = [run(lo, min(lo + 100, T)) for lo in range(0, T, 100)]
    phi_c, f_c, ne_c = (torch.cat([p[i] for p in parts]) for i in range(3))
    assert torch.equal(ne_all, ne_c)
    assert (f_all - f_c).abs().max().item() <= 1e-6 * f_c.abs().max().item(), (f_all - f_c).abs().max().item()
    assert (phi_all - phi_c).abs().max().item() <= 1e-5, (phi_all - phi_c).abs().max().item()
    print("two-tasks-per-CU fit vs resident: max |d f| %.2e, max |d phi| %.2e" % ((f_all - f_c).abs().max().item(), (phi_all - phi_c).abs().max().item()))


def test_fit_in_place_is_the_same_fit(dev):
    """``gp_ops.fit(..., inplace=True)`` (what the meta-step uses for its freshly initialised parameters: one copy kernel
    less per step) runs the same optimisation as the copying default and leaves the result in ``phi0`` itself."""
    from adkf_ift_amd import gp_ops
    from adkf_ift_amd.synthetic import make_tasks

    T, N, d = 8, 48, 24
    tasks = make_tasks(T, N, d, first_task=90)
    Zs, _ = tasks.features()
    pri = torch.empty(T, 4, device=dev)
    b = gp_ops.GPBatch(Zs.to(dev), tasks.y_s.to(dev), pri, "rbf")
    phi0, _ = gp_ops.init_params_batch(b)
    b.flags = gp_ops.REUSE_DIST
    keep = phi0.clone()
    phi_a, f_a, _, ne_a, info = gp_ops.fit(b, phi0, max_evals=40)
    gp_ops.check_info(info)
    assert torch.equal(phi0, keep) and phi_a.data_ptr() != phi0.data_ptr()
    phi_b, f_b, _, ne_b, info = gp_ops.fit(b, phi0, max_evals=40, inplace=True)
    gp_ops.check_info(info)
    assert phi_b.data_ptr() == phi0.data_ptr()
    assert torch.equal(phi_a, phi_b) and torch.equal(f_a, f_b) and torch.equal(ne_a, ne_b)
