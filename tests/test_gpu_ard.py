"""GPU: the ARD kernel (one lengthscale per feature dimension, h = 2 + d) through the C ABI against the autograd fixtures
(tests/golden/ard_*.npz): value and gradient of f_in, predictive mean / variance, f_out and its gradient in the h raw
parameters, the conjugate-gradient solution v of H v = grad f_out, the total IFT gradient dL/dZ, a Hessian-vector product
against the stored dense H, and the L-BFGS inner fit against the float64 oracle optimum."""
import glob
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-4


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _batch(z, dev, pad_s=0, pad_q=0):
    from adkf_ift_amd import gp_ops

    n, m, d = z["Z_s"].shape[0], z["Z_q"].shape[0], z["Z_s"].shape[1]
    f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
    Zs, Zq = torch.zeros(1, n + pad_s, d, device=dev), torch.zeros(1, m + pad_q, d, device=dev)
    ys, yq = torch.zeros(1, n + pad_s, device=dev), torch.zeros(1, m + pad_q, device=dev)
    Zs[0, :n], Zq[0, :m], ys[0, :n], yq[0, :m] = f32(z["Z_s"]), f32(z["Z_q"]), f32(z["y_s"]), f32(z["y_q"])
    if pad_s:
        Zs[0, n:] = 7.0   # garbage in the padding must not matter
    kw = dict(n_s=torch.tensor([n]), n_q=torch.tensor([m])) if (pad_s or pad_q) else {}
    b = gp_ops.GPBatch(Zs, ys, f32(z["priors"])[None], int(z["kind"]), Z_q=Zq, y_q=yq, ard=True, **kw)
    return b, f32(z["phi"])[None], n, m


@pytest.mark.parametrize("pad", [(0, 0), (4, 8)])
def test_ard_golden_cases(golden_dir, dev, pad):
    from adkf_ift_amd import gp_ops

    files = sorted(glob.glob(os.path.join(golden_dir, "ard_*.npz")))
    assert len(files) >= 5
    worst = {}
    for f in files:
        z = np.load(f)
        b, phi, n, m = _batch(z, dev, *pad)
        fin, g, dZ, info = gp_ops.mll_value_grad(b, phi, want_dZ=True)
        gp_ops.check_info(info)
        mean, var, _, info = gp_ops.predict(b, phi)
        gp_ops.check_info(info)
        out = gp_ops.ift_hypergrad(b, phi, cg_maxiter=64, cg_tol=1e-7)
        gp_ops.check_info(out["info"])
        got = {"f_in": fin[0].item(), "g_in": g[0].cpu().numpy(), "dfin_dZs": dZ[0, :n].cpu().numpy(),
               "pred_mean": mean[0, :m].cpu().numpy(), "pred_var": var[0, :m].cpu().numpy(), "f_out": out["f_out"][0].item(),
               "g_out": out["g_phi"][0].cpu().numpy(), "v": out["v"][0].cpu().numpy(),
               "dZs_total": out["dZ_s"][0, :n].cpu().numpy(), "dZq_total": out["dZ_q"][0, :m].cpu().numpy()}
        for k, v in got.items():
            e = rel(v, z[k])
            if k == "g_in":   # vanishes at a fitted point: measure against the O(1e-2) scale of an unfitted gradient
                e = np.abs(np.asarray(v, dtype=np.float64) - z[k]).max() / max(np.abs(z[k]).max(), 1e-2)
            worst[k] = max(worst.get(k, 0.0), e)
            # v = H^-1 g_out in float32 cannot be better than eps32 cond(H): H has 2 + d rows here and cond(H) = 1.6e3 for the
            # 128-point fixture (10 .. 25 for the others), where the v-dependent term is 88 % of dL/dZ_s.  Those two outputs are
            # held to max(1e-4, 4 eps32 cond(H)) with cond(H) taken from the fixture's own dense Hessian (3.9e-4 for that
            # fixture, 1e-4 for the rest; observed 1.0e-4 .. 1.1e-4 depending on the summation order of the distance kernel);
            # everything else to 1e-4.
            tol = max(TOL, 4 * 6e-8 * float(np.linalg.cond(z["H"]))) if k in ("v", "dZs_total") else TOL
            assert e <= tol, (os.path.basename(f), k, e, int(out["cg_iters"][0]))
        if pad[0]:
            assert float(out["dZ_s"][0, n:].abs().max()) == 0.0 and float(out["dZ_q"][0, m:].abs().max()) == 0.0
        # first-order flag: no CG, dZ = direct part
        o1 = gp_ops.ift_hypergrad(b, phi, ignore_grad_correction=True)
        assert rel(o1["dZ_s"][0, :n].cpu().numpy(), z["dZs_total"] + z["mixed_Zs"]) <= TOL
        assert int(o1["cg_iters"][0]) == 0
    print("worst ARD errors:", {k: f"{v:.1e}" for k, v in worst.items()})


def test_ard_batched_ragged_equals_single(golden_dir, dev):
    """Three ARD tasks of different sizes in one padded batch == each alone."""
    from adkf_ift_amd import gp_ops

    names = ["ard_N16_Nq24_d12_k1_r0_s1", "ard_N8_Nq8_d4_k0_r0_s0"]
    z0 = np.load(os.path.join(golden_dir, names[0] + ".npz"))
    # same d is required inside one batch: build a second task by truncating the first
    n2, m2, d = 9, 11, 12
    f32 = lambda a: torch.tensor(np.asarray(a), dtype=torch.float32, device=dev)
    Zs, Zq = torch.zeros(2, 16, d, device=dev), torch.zeros(2, 24, d, device=dev)
    ys, yq = torch.zeros(2, 16, device=dev), torch.zeros(2, 24, device=dev)
    Zs[0], Zq[0], ys[0], yq[0] = f32(z0["Z_s"]), f32(z0["Z_q"]), f32(z0["y_s"]), f32(z0["y_q"])
    Zs[1, :n2], Zq[1, :m2], ys[1, :n2], yq[1, :m2] = Zs[0, :n2] * 1.1, Zq[0, :m2] * 0.9, ys[0, :n2], yq[0, :m2]
    pri = f32(z0["priors"])[None].repeat(2, 1)
    phi = f32(z0["phi"])[None].repeat(2, 1)
    phi[1] += 0.1
    both = gp_ops.GPBatch(Zs, ys, pri, 1, Z_q=Zq, y_q=yq, n_s=torch.tensor([16, n2]), n_q=torch.tensor([24, m2]), ard=True)
    ob = gp_ops.ift_hypergrad(both, phi)
    one = gp_ops.GPBatch(Zs[1:, :n2].contiguous(), ys[1:, :n2].contiguous(), pri[1:], 1, Z_q=Zq[1:, :m2].contiguous(),
                         y_q=yq[1:, :m2].contiguous(), ard=True)
    o1 = gp_ops.ift_hypergrad(one, phi[1:])
    assert rel(ob["f_out"][1].item(), o1["f_out"][0].item()) <= 1e-5
    assert rel(ob["v"][1].cpu().numpy(), o1["v"][0].cpu().numpy()) <= 1e-3
    assert rel(ob["dZ_s"][1, :n2].cpu().numpy(), o1["dZ_s"][0].cpu().numpy()) <= 1e-4


def test_ard_fit_reaches_oracle_optimum(golden_dir, dev):
    from adkf_ift_amd import gp_ops
    from oracle import gp_oracle as O

    for name in ("ard_N32_Nq32_d16_k0_r1_s0", "ard_N48_Nq40_d24_k1_r0_s2"):
        z = np.load(os.path.join(golden_dir, name + ".npz"))
        b, _, n, m = _batch(z, dev)
        b.priors = torch.empty(1, 4, device=dev)
        phi0, l0 = gp_ops.init_params_batch(b, use_numeric_labels=bool(z["regression"]))
        assert rel(phi0[0].cpu().numpy(), z["phi0"]) <= 1e-5 and rel(b.priors[0].cpu().numpy(), z["priors"]) <= 1e-5
        phi, f, gn, ne, info = gp_ops.fit(b, phi0, max_evals=300)
        gp_ops.check_info(info)
        pri = O.Priors(*[float(v) for v in z["priors"]])
        Zs, ys = torch.tensor(z["Z_s"]).double(), torch.tensor(z["y_s"]).double()
        f_ref = O.f_inner(Zs, ys, torch.tensor(z["phi"]), pri, int(z["kind"])).item()    # the fixture's phi is the SciPy optimum
        f_got = O.f_inner(Zs, ys, phi[0].double().cpu(), pri, int(z["kind"])).item()
        assert abs(f[0].item() - f_got) <= 1e-4 * abs(f_got)
        assert f_got <= f_ref + 2e-5 * abs(f_ref), (name, f_got, f_ref, int(ne[0]))
        assert gn[0].item() <= 1e-3 and int(ne[0]) <= 300
        # the REUSE_INNER hand-over: hypergradient right after the fit == hypergradient from scratch
        b.flags = gp_ops.REUSE_INNER
        o_reuse = gp_ops.ift_hypergrad(b, phi)
        b.flags = 0
        o_fresh = gp_ops.ift_hypergrad(b, phi)
        assert rel(o_reuse["dZ_s"][0].cpu().numpy(), o_fresh["dZ_s"][0].cpu().numpy()) <= 1e-5
        q = O.full_reference_quantities(Zs, ys, torch.tensor(z["Z_q"]).double(), torch.tensor(z["y_q"]).double(), phi[0].double().cpu(), pri, int(z["kind"]))
        assert rel(o_fresh["dZ_s"][0].cpu().numpy(), q["dZs_total"]) <= TOL
        assert rel(o_fresh["dZ_q"][0].cpu().numpy(), q["dZq_total"]) <= TOL


def test_ard_model_surface_and_batched_meta_step(dev):
    """``use_ard=True`` through the reference-shaped surface: reinit (every lengthscale at the median heuristic, [1, d]
    parameter), fit, fused cauchy_hypergradient == the dense reference algorithm on the float64 oracle; and the batched
    meta-step equals the per-task loop."""
    from adkf_ift_amd.hypergradient import cauchy_hypergradient
    from adkf_ift_amd.models import ADKTModel, ADKTModelConfig, fit_gpytorch_scipy
    from adkf_ift_amd.trainer import MetaStepConfig, meta_step
    from oracle import gp_oracle as O
    from oracle.hypergrad_oracle import dense_ift_hypergradient
    from test_gpu_surface import make_batch

    torch.manual_seed(3)
    cfg = ADKTModelConfig(used_features="ecfp+fc", gp_kernel="matern", use_ard=True, fc_hidden_dim=16, fc_out_dim=8)
    model = ADKTModel(cfg).to(dev)
    batches = [make_batch(dev, ns=16, nq=24, seed=s) for s in (11, 12)]
    model.train()
    acc = [torch.zeros_like(p) for p in model.feature_extractor_params()]
    vals = []
    for bi, batch in enumerate(batches):
        assert model(batch, train_loss=True) is None
        assert model.gp_model.covar_module.base_kernel.raw_lengthscale.shape == (1, 8)
        fit_gpytorch_scipy(model.mll)
        f_outer, f_inner = model.task_losses(batch)
        po, pi = tuple(model.feature_extractor_params()), tuple(model.gp_params())
        val = cauchy_hypergradient(f_outer, f_inner, po, pi, dev)
        vals.append(val.item() / 24)
        for a, p in zip(acc, po):
            a += p.grad / len(batches)
        if bi == 0:   # against the dense algorithm (h = 10: Hessian and mixed Jacobian by autograd) in float64
            Xs, Xq = batch.support_features.fingerprints.double().cpu(), batch.query_features.fingerprints.double().cpu()
            ys, yq = (batch.support_labels.double().cpu() - 0.5) * 2, (batch.query_labels.double().cpu() - 0.5) * 2
            pri = O.Priors(*model.mll.priors_row(torch.device("cpu"))[0].double().tolist())
            feats = lambda p, X: torch.relu(X @ p[0].T + p[1]) @ p[2].T + p[3]
            fin = lambda p, q: O.f_inner(feats(p, Xs), ys, torch.cat([t.reshape(-1) for t in q]), pri, 1)
            fout = lambda p, q: O.f_outer(feats(p, Xs), ys, feats(p, Xq), yq, torch.cat([t.reshape(-1) for t in q]), 1)
            p64 = tuple(p.detach().double().cpu().requires_grad_() for p in po)
            q64 = tuple(p.detach().double().cpu().requires_grad_() for p in pi)
            ref_val = dense_ift_hypergradient(fout, fin, p64, q64)
            assert abs(val.item() - ref_val.item()) <= 1e-4 * abs(ref_val.item())
            scale = max(q.grad.abs().max().item() for q in p64)
            for p, q in zip(po, p64):
                assert (p.grad.double().cpu() - q.grad).abs().max().item() <= 1e-3 * scale
    # batched: both tasks in one library call, one backward
    for p in model.parameters():
        p.grad = None
    X_s = torch.stack([b.support_features.fingerprints for b in batches])
    X_q = torch.stack([b.query_features.fingerprints for b in batches])
    y_s = torch.stack([(b.support_labels.float() - 0.5) * 2 for b in batches])
    y_q = torch.stack([(b.query_labels.float() - 0.5) * 2 for b in batches])
    params = list(model.feature_extractor_params())
    mcfg = MetaStepConfig(gp_kernel="matern", use_ard=True, clip_value=None)
    losses, _ = meta_step(lambda: (model.fc(X_s), model.fc(X_q)), params, None, y_s, y_q, mcfg, check=True)
    scale = max(a.abs().max().item() for a in acc)
    for a, p in zip(acc, params):
        assert (p.grad - a).abs().max().item() <= 2e-3 * scale
    assert np.abs(losses.cpu().numpy() - np.array(vals)).max() <= 1e-3 * np.abs(vals).max()
