"""CPU: pins the oracle itself.  (i) the autograd oracle reproduces the committed golden vectors (guards against
drift of oracle/gp_oracle.py after the fixtures were generated); (ii) the closed-form staged restatement
(oracle/closed_form.py, the algebra the HIP kernels implement) agrees with them; (iii) GPyTorch-semantics
spot checks that are derivable by hand."""
import glob
import math
import os

import numpy as np
import torch

from oracle import closed_form as C
from oracle import gp_oracle as O


def rel(a, ref):
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-300)


def test_closed_form_matches_golden(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "gp_*.npz")))
    assert len(files) >= 30
    for f in files:
        g = np.load(f)
        out = C.full_pipeline(g["Z_s"], g["y_s"], g["Z_q"], g["y_q"], g["phi"], g["priors"], int(g["kind"]))
        for k, v in out.items():
            if k in g.files:
                tol = 1e-6 if g[k].dtype == np.float32 else 1e-8
                if k == "g_in":
                    assert np.abs(np.asarray(v) - g[k]).max() <= 1e-8, (f, k)
                else:
                    assert rel(v, g[k]) <= tol, (os.path.basename(f), k, rel(v, g[k]))


def test_autograd_oracle_reproduces_small_golden(golden_dir):
    for name in ("gp_N8_Nq8_d4_k0_r0_s1", "gp_N16_Nq32_d16_k1_r1_s1", "gp_N5_Nq3_d7_k0_r0_s2"):
        g = np.load(os.path.join(golden_dir, name + ".npz"))
        pri = O.Priors(*g["priors"].tolist())
        q = O.full_reference_quantities(torch.tensor(g["Z_s"]), torch.tensor(g["y_s"]), torch.tensor(g["Z_q"]),
                                        torch.tensor(g["y_q"]), torch.tensor(g["phi"]), pri, int(g["kind"]))
        for k in ("f_in", "g_in", "H", "f_out", "g_out", "v", "dZs_total", "dZq_total", "pred_mean", "pred_var", "l0"):
            assert rel(q[k], g[k]) <= 1e-9, (name, k)


def test_gpytorch_semantics_by_hand():
    # A1: softplus parametrisation with the 1e-4 noise floor, raw_outputscale = 0 -> ln 2
    phi = torch.tensor([0.3, 0.0, -0.2], dtype=torch.float64)
    noise, os_, ls = O.transform_phi(phi)
    assert abs(noise.item() - (math.log1p(math.exp(0.3)) + 1e-4)) < 1e-12
    assert abs(os_.item() - math.log(2.0)) < 1e-12
    # inverse transform used by `.noise = 0.1` / `.lengthscale = l0`
    assert abs(torch.nn.functional.softplus(O.inv_softplus(0.37)).item() - 0.37) < 1e-12
    # A4: LogNormal(loc = log(mode) + s^2, s) has its mode at `mode`
    loc, sc = O.noise_prior_params(False)
    xs = torch.linspace(0.05, 0.2, 3001, dtype=torch.float64)
    lp = torch.stack([O.lognormal_log_prob(x, loc, sc) for x in xs])
    assert abs(xs[lp.argmax()].item() - 0.1) < 1e-4
    # A3: kernels at zero distance equal the outputscale; Matern-5/2 closed form at r = 1
    Z = torch.zeros(2, 3, dtype=torch.float64)
    Z[1, 0] = 2.0
    one, two = torch.tensor(1.0, dtype=torch.float64), torch.tensor([2.0], dtype=torch.float64)
    K = O.kernel_matrix(Z, Z, 1.7 * one, two, O.KERNEL_MATERN52)
    assert abs(K[0, 0].item() - 1.7) < 1e-12
    r = 1.0
    assert abs(K[0, 1].item() - 1.7 * (1 + math.sqrt(5) * r + 5.0 / 3.0 * r * r) * math.exp(-math.sqrt(5) * r)) < 1e-12
    K = O.kernel_matrix(Z, Z, 1.7 * one, two, O.KERNEL_RBF)
    assert abs(K[0, 1].item() - 1.7 * math.exp(-0.5)) < 1e-12
    # a3: torch.median is the LOWER median, over strictly-positive upper-triangle entries
    Z = torch.tensor([[0.0], [1.0], [3.0], [3.0]], dtype=torch.float64)   # d2 in {1, 9, 9, 4, 4, 0}: positives 1,4,4,9,9
    assert abs(O.median_lengthscale_init(Z).item() - math.sqrt(0.5 * 4.0)) < 1e-12
    Z = torch.tensor([[0.0], [1.0], [3.0]], dtype=torch.float64)          # d2 in {1, 9, 4}; lower median of 3 = 4
    assert abs(O.median_lengthscale_init(Z).item() - math.sqrt(0.5 * 4.0)) < 1e-12
    Z = torch.tensor([[0.0], [1.0]], dtype=torch.float64)
    assert abs(O.median_lengthscale_init(Z).item() - math.sqrt(0.5)) < 1e-12


def test_mll_is_divided_by_n_after_priors():
    torch.manual_seed(0)
    Z = torch.randn(6, 3, dtype=torch.float64)
    y = torch.randn(6, dtype=torch.float64)
    phi = torch.tensor([0.1, 0.2, 0.3], dtype=torch.float64)
    pri = O.Priors(-2.2, 0.25, 0.4, 0.25)
    noise, os_, ls = O.transform_phi(phi)
    A = O.kernel_matrix(Z, Z, os_, ls, 0) + noise * torch.eye(6, dtype=torch.float64)
    mvn = torch.distributions.MultivariateNormal(torch.zeros(6, dtype=torch.float64), A).log_prob(y)
    expect = -(mvn + O.lognormal_log_prob(noise, -2.2, 0.25) + O.lognormal_log_prob(ls, 0.4, 0.25)) / 6
    assert abs(O.f_inner(Z, y, phi, pri, 0).item() - expect.item()) < 1e-12


def test_ard_closed_forms_match_autograd_fixtures(golden_dir):
    """The scaled-feature / homogeneity formulation of the ARD kernel (oracle/closed_form_ard.py: gradient, Hessian-vector
    product, mixed term, CG solve) against the autograd fixtures."""
    import glob
    from oracle.closed_form_ard import ArdTask, cg_solve, full_pipeline_ard

    files = sorted(glob.glob(os.path.join(golden_dir, "ard_*.npz")))
    assert len(files) >= 5
    for f in files:
        z = np.load(f)
        dense = z["Z_s"].shape[0] <= 48
        r = full_pipeline_ard(z["Z_s"], z["y_s"], z["Z_q"], z["y_q"], z["phi"], z["priors"], int(z["kind"]), dense_solve=dense)
        keys = ["f_in", "g_in", "f_out", "g_out", "v", "dfin_dZs", "mixed_Zs", "dZs_total", "dZq_total", "pred_mean", "pred_var"]
        if dense:
            keys.append("H")
        for k in keys:
            want = np.asarray(z[k], dtype=np.float64)
            err = np.abs(np.asarray(r[k]) - want).max() / max(np.abs(want).max(), 1e-300)
            assert err <= (1e-9 if dense else 1e-6), (os.path.basename(f), k, err)
    # Hessian-vector product against a column combination of the stored H; CG recovers v
    z = np.load(files[0])
    t = ArdTask(z["Z_s"], z["y_s"], z["phi"], z["priors"], int(z["kind"]))
    u = np.linspace(-1, 1, z["phi"].size)
    assert np.abs(t.hvp(u) - z["H"] @ u).max() <= 1e-10 * np.abs(z["H"] @ u).max()
    v, its = cg_solve(t.hvp, z["g_out"])
    assert np.abs(v - z["v"]).max() <= 1e-8 * np.abs(z["v"]).max() and its <= 3 * z["phi"].size


def test_oracle_against_independent_published_implementations():
    """The oracle's GP formulas are a restatement of GPyTorch from memory (parity unpinned at that boundary).  Two
    independent, published implementations available in this image pin the arithmetic that does NOT depend on GPyTorch's
    parametrisation: scikit-learn's exact GP (log marginal likelihood, posterior mean and covariance, RBF and Matern-5/2
    kernels) and torch.distributions.LogNormal (the prior density).  Still memory-only after this: the softplus
    parametrisation, the 1e-4 noise floor, priors added before the division by N (oracle/gp_oracle.py header)."""
    from sklearn.gaussian_process import GaussianProcessRegressor
    from sklearn.gaussian_process.kernels import RBF, ConstantKernel, Matern, WhiteKernel

    torch.manual_seed(3)
    N, M, d = 24, 17, 5
    Zs, Zq = torch.randn(N, d, dtype=torch.float64), torch.randn(M, d, dtype=torch.float64)
    ys = torch.randn(N, dtype=torch.float64)
    phi = torch.tensor([-1.3, 0.4, 1.1], dtype=torch.float64)
    noise, os_, ls = (t.item() for t in O.transform_phi(phi))
    for kind, base in ((O.KERNEL_RBF, RBF(length_scale=ls)), (O.KERNEL_MATERN52, Matern(length_scale=ls, nu=2.5))):
        kern = ConstantKernel(os_) * base + WhiteKernel(noise)
        # kernel matrices
        K_sk = (ConstantKernel(os_) * base)(Zs.numpy(), Zq.numpy())
        K_or = O.kernel_matrix(Zs, Zq, torch.tensor(os_, dtype=torch.float64), torch.tensor([ls], dtype=torch.float64), kind).numpy()
        assert rel(K_or, K_sk) <= 1e-12
        gpr = GaussianProcessRegressor(kernel=kern, alpha=0.0, optimizer=None, normalize_y=False).fit(Zs.numpy(), ys.numpy())
        # log marginal likelihood: f_inner without priors = -lml / N
        lml = gpr.log_marginal_likelihood(gpr.kernel_.theta)
        # (noise prior scale <= 0 is not a supported oracle configuration: subtract the prior term explicitly instead)
        pri = O.Priors(*O.noise_prior_params(False))
        f = O.f_inner(Zs, ys, phi, pri, kind).item()
        lp = O.lognormal_log_prob(torch.tensor(noise, dtype=torch.float64), pri.noise_loc, pri.noise_scale).item()
        assert abs((-f * N - lp) - lml) <= 1e-10 * abs(lml)
        # posterior: sklearn's predict() is the latent posterior; the oracle adds the likelihood noise (App. A6)
        mean_sk, cov_sk = gpr.predict(Zq.numpy(), return_cov=True)
        mean_or, cov_or = O.predict(Zs, ys, Zq, phi, kind)
        assert rel(mean_or.numpy(), mean_sk) <= 1e-9
        # WhiteKernel contributes its noise to k(x*, x*) in sklearn, i.e. the same "with likelihood noise" covariance
        assert rel(cov_or.numpy(), cov_sk) <= 1e-8
    # LogNormal prior density (gpytorch.priors.LogNormalPrior is a TransformedDistribution(Normal, Exp))
    x = torch.tensor([0.03, 0.1, 0.7, 2.5], dtype=torch.float64)
    for loc, sc in ((-2.24, 0.25), (0.8, 0.25), (0.0, 1.3)):
        ref = torch.distributions.LogNormal(torch.tensor(loc, dtype=torch.float64), torch.tensor(sc, dtype=torch.float64)).log_prob(x)
        got = torch.stack([O.lognormal_log_prob(xi, loc, sc) for xi in x])
        assert (got - ref).abs().max().item() <= 1e-12
        assert abs(O.lognormal_log_prob(x, loc, sc).item() - ref.sum().item()) <= 1e-11   # ARD: summed over elements
    # median heuristic: torch.median is the LOWER median of the strictly-upper-triangular positive squared distances
    Z = torch.tensor([[0.0], [1.0], [3.0], [7.0]], dtype=torch.float64)   # d^2: 1, 9, 49, 4, 36, 16 -> sorted 1 4 9 16 36 49 -> lower median 9
    assert abs(O.median_lengthscale_init(Z).item() - math.sqrt(0.5 * 9.0)) < 1e-12
