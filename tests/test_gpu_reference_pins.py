"""GPU: the fused HIP hypergradient against numbers that came out of the REFERENCE's own operators.

``tests/golden/linmap_*.npz`` and ``harness_*.npz`` hold theta.grad as computed by /root/reference's
``fs_mol/utils/cauchy_hypergradient.py:5-163`` and ``cauchy_hypergradient_jvp.py:118-129`` (run in the build container by
``tests/golden/make_golden.py``; nothing here reads the reference tree).  The device path is evaluated AT THE FIXTURE'S
phi - no inner fit - so the comparison measures the operator, not the optimiser, and is held to the north-star
tolerance: 1e-4 relative (max-norm against the largest entry of the expected array).  The fitted variants (device
optimiser + operator) are separate tests with their own, looser, bound.
"""
import math
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 1e-4


def rel(a, ref):
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30)


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    return torch.device("cuda:0")


def _linmap_inputs(g):
    """(X_s, X_q, y_s, y_q, W) float32 CPU tensors of a linmap fixture (stored, or regenerated from its seed)."""
    if "X_s" in g.files:
        return tuple(torch.tensor(g[k]) for k in ("X_s", "X_q", "y_s", "y_q", "W"))
    from adkf_ift_amd.synthetic import make_tasks
    t = make_tasks(1, int(g["N"]), int(g["d"]), N_q=int(g["Nq"]), regression=False, first_task=int(g["seed"]))
    return t.X_s[0], t.X_q[0], t.y_s[0], t.y_q[0], t.W


@pytest.mark.parametrize("name", ["linmap_N16_Nq24_d12_k0", "linmap_N16_Nq24_d12_k1",
                                  "linmap_N128_Nq128_d256_k0", "linmap_N128_Nq128_d256_k1"])
def test_fused_hypergradient_matches_reference_run(golden_dir, dev, name):
    from adkf_ift_amd import gp_ops
    from adkf_ift_amd.synthetic import LinearFeatureMap

    g = np.load(os.path.join(golden_dir, name + ".npz"))
    X_s, X_q, y_s, y_q, W0 = (a.to(dev) for a in _linmap_inputs(g))
    pri = torch.tensor(g["priors"], dtype=torch.float32)[None].to(dev)
    phi = torch.tensor(g["phi"], dtype=torch.float32)[None].to(dev)

    def theta_grad(**flags):
        W = W0.clone().requires_grad_(True)
        feats = LinearFeatureMap(X_s[None], X_q[None], W, chunks=1)()
        Z_s, Z_q = feats[0], feats[1]
        b = gp_ops.GPBatch(Z_s.detach(), y_s[None], pri, int(g["kind"]), Z_q=Z_q.detach(), y_q=y_q[None])
        out = gp_ops.ift_hypergrad(b, phi, **flags)
        gp_ops.check_info(out["info"])
        torch.autograd.backward([Z_s, Z_q], [out["dZ_s"], out["dZ_q"]])
        return W.grad.cpu().numpy(), out

    gW, out = theta_grad()
    assert rel(out["f_out"][0].item(), g["f_out"]) <= TOL
    assert rel(gW, g["grad_W_dense"]) <= TOL, rel(gW, g["grad_W_dense"])
    if "grad_W_jvp" in g.files:
        assert rel(gW, g["grad_W_jvp"]) <= TOL
    assert rel(out["g_phi"][0].cpu().numpy(), g["grad_phi"]) <= TOL      # phi.grad = d f_out / d phi (cauchy_hypergradient.py:139-163)
    gW1, _ = theta_grad(ignore_grad_correction=True)
    assert rel(gW1, g["grad_W_first_order"]) <= TOL
    print(name, "theta.grad rel err %.2e (first order %.2e)" % (rel(gW, g["grad_W_dense"]), rel(gW1, g["grad_W_first_order"])))


class FixedPhiBackend:
    """HipGPBackend without the inner fit: re-initialisation (for the priors) then the IFT hypergradient at a given phi.
    Every number still comes from libadkf_gp.so."""

    def __init__(self, phi):
        self.phi = phi

    def run(self, Z_s, y_s, Z_q, y_q, cfg, n_s=None, n_q=None, fit_events=None, out_dZ=None):
        from adkf_ift_amd import gp_ops
        priors = torch.empty(Z_s.shape[0], 4, dtype=torch.float32, device=Z_s.device)
        b = gp_ops.GPBatch(Z_s, y_s, priors, cfg.gp_kernel, Z_q=Z_q, y_q=y_q, n_s=n_s, n_q=n_q)
        gp_ops.init_params_batch(b, cfg.use_numeric_labels, cfg.use_lengthscale_prior)
        b.flags = gp_ops.REUSE_DIST
        out = gp_ops.ift_hypergrad(b, self.phi, ignore_grad_correction=cfg.ignore_grad_correction, out_dZ=out_dZ)
        zero = torch.zeros_like(out["info"])
        self.priors = priors
        return self.phi, out["f_out"], out["dZ_s"], out["dZ_q"], zero, out["info"]


@pytest.mark.parametrize("name", ["harness_T4_N16_d8_k0", "harness_T4_N128_d256_k0"])
def test_meta_step_at_fixture_phi_matches_reference_loop(golden_dir, dev, name):
    """Row H (fs_mol/utils/adaptive_dkt_utils.py:352-413): task-mean of the hypergradients, clip-by-global-norm, step."""
    from adkf_ift_amd.synthetic import make_tasks
    from adkf_ift_amd.trainer import MetaStepConfig, meta_step

    g = np.load(os.path.join(golden_dir, name + ".npz"))
    T, N, d = int(g["T"]), int(g["N"]), int(g["d"])
    tasks = make_tasks(T, N, d, first_task=500).to(dev)
    W = tasks.W.clone().requires_grad_(True)
    opt = torch.optim.SGD([W], lr=0.5)
    feats = lambda: (tasks.X_s @ W / math.sqrt(d), tasks.X_q @ W / math.sqrt(d))
    W0 = W.detach().clone()
    backend = FixedPhiBackend(torch.tensor(g["phi"], dtype=torch.float32).to(dev))
    losses, _ = meta_step(feats, [W], opt, tasks.y_s, tasks.y_q, MetaStepConfig(gp_kernel="rbf", clip_value=1.0),
                          backend=backend, check=True)
    assert rel(backend.priors.cpu().numpy(), g["priors"]) <= 1e-5           # a3/a4 on the device == the fixture's priors
    assert rel(W.grad.cpu().numpy(), g["grad_clipped"]) <= TOL, rel(W.grad.cpu().numpy(), g["grad_clipped"])
    assert rel((W0 - W.detach()).cpu().numpy() / 0.5, g["grad_clipped"]) <= TOL
    assert rel(losses.cpu().numpy() * N, g["f_out"]) <= TOL
    # the un-clipped mean as well (the SGD step above moved W: back to the start point first)
    with torch.no_grad():
        W.copy_(W0)
    W.grad = None
    meta_step(feats, [W], None, tasks.y_s, tasks.y_q, MetaStepConfig(gp_kernel="rbf", clip_value=None), backend=backend)
    assert rel(W.grad.cpu().numpy(), g["grad_mean"]) <= TOL
    assert abs(float(W.grad.norm()) - float(g["grad_norm"])) <= TOL * float(g["grad_norm"])


@pytest.mark.parametrize("name", ["harness_T4_N16_d8_k0", "harness_T4_N128_d256_k0"])
def test_meta_step_with_device_fit(golden_dir, dev, name):
    """The optimiser test: the same step with the DEVICE inner fit (fp32 BFGS) against the fixture whose phi came from a
    float64 L-BFGS-B.  The two optima differ at the 1e-3 level in raw phi (flat lengthscale valley), which moves the
    hypergradient by more than fp32 arithmetic does - hence the looser bound here and the fixed-phi test above."""
    from adkf_ift_amd.synthetic import make_tasks
    from adkf_ift_amd.trainer import MetaStepConfig, meta_step

    g = np.load(os.path.join(golden_dir, name + ".npz"))
    T, N, d = int(g["T"]), int(g["N"]), int(g["d"])
    tasks = make_tasks(T, N, d, first_task=500).to(dev)
    W = tasks.W.clone().requires_grad_(True)
    feats = lambda: (tasks.X_s @ W / math.sqrt(d), tasks.X_q @ W / math.sqrt(d))
    losses, phi = meta_step(feats, [W], None, tasks.y_s, tasks.y_q, MetaStepConfig(gp_kernel="rbf", clip_value=1.0), check=True)
    # Same optimum?  Judged on the OBJECTIVE, not on the distance in phi: along the flat lengthscale valley a 1e-3 relative
    # move of l changes f_inner by less than a float32 ulp, so where exactly a float32 fit stops there depends on the rounding
    # of its reductions (seen: |d phi_l| = 1e-3 and 1.6e-2 for the same f).  The device optimum must be at least as good as
    # the reference's under the device's own evaluation, stationary to float32 resolution (test_fit_reaches_oracle_optimum
    # explains the 5e-4), and in the same basin.
    from adkf_ift_amd import gp_ops
    with torch.no_grad():
        Zs, _ = feats()
    _, pri, _ = gp_ops.init_params(Zs)
    b = gp_ops.GPBatch(Zs, tasks.y_s, pri, "rbf")
    phi_ref = torch.tensor(g["phi"], dtype=torch.float32).to(dev)
    f_dev, g_dev, _, _ = gp_ops.mll_value_grad(b, phi)
    f_ref, _, _, _ = gp_ops.mll_value_grad(b, phi_ref)
    assert (f_dev <= f_ref + 2e-7 * f_ref.abs() + 1e-7).all(), (f_dev, f_ref)
    assert g_dev.abs().max().item() <= 5e-4
    assert np.abs(phi.cpu().numpy() - g["phi"]).max() <= 5e-2
    assert rel(W.grad.cpu().numpy(), g["grad_clipped"]) <= 2e-3
    assert rel(losses.cpu().numpy() * N, g["f_out"]) <= 1e-3
