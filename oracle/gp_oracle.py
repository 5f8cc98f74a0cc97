"""CPU oracle for the ADKF-IFT GP hot path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this module.  The product path (``adkf_ift_amd``) never routes through it and
fails loudly when the HIP extension is missing.

PARITY UNPINNED at the GPyTorch boundary: the reference (Wenlin-Chen/ADKF-IFT) does all
GP arithmetic inside GPyTorch/BoTorch, which are neither vendored in ``/root/reference``
nor installed here, and no reference test pins an MLL value, a Cholesky, a predictive
mean or a GP hypergradient.  This file therefore *restates*
the published GPyTorch formulas the reference calls (gpytorch 1.6-1.8 era, implied by
``environment.yml:9,19-22``), in float64 torch so that autograd supplies every
derivative independently of the closed forms the HIP kernels implement.  What IS pinned:

* the GP arithmetic that does not depend on GPyTorch's parametrisation - kernel matrices (RBF, Matern-5/2), the log
  marginal likelihood, the exact posterior mean/covariance with likelihood noise - by scikit-learn's exact GP, and the
  LogNormal prior density by torch.distributions (tests/test_oracle.py::test_oracle_against_independent_...).
  Still memory-only after that: softplus parametrisation of noise / outputscale / lengthscale (A1),
  the 1e-4 noise floor (A1), prior log-densities evaluated on the transformed values and added BEFORE the division by N
  (A4), fresh raw_outputscale = 0 (A1), GPyTorch centring its kernel inputs (A3; immaterial for stationary kernels);

* the hypergradient operator, by the reference's own ``cauchy_hypergradient`` /
  ``cauchy_hypergradient_jvp`` (torch-only files, imported by path inside this
  container when generating ``tests/golden``), and by the known answers of
  ``/root/reference/test_hypergrad.ipynb`` (cells 5-9, 16-25, 29-30);
* the model structure / modes / priors / initialisation, by the reference source lines
  cited on each function below.

Every function cites the reference file:line it follows (paths relative to
``/root/reference``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch
import torch.nn.functional as F

DT = torch.float64
NOISE_LOWER_BOUND = 1e-4  # gpytorch GaussianLikelihood default GreaterThan(1e-4)
LOG_2PI = math.log(2.0 * math.pi)

KERNEL_RBF = 0
KERNEL_MATERN52 = 1


# --------------------------------------------------------------------------------------
# parameter transforms (gpytorch Positive()/GreaterThan() constraints = softplus; App. A1)
# --------------------------------------------------------------------------------------
def softplus(x: torch.Tensor) -> torch.Tensor:
    return F.softplus(x)


def inv_softplus(y):
    """Inverse of softplus, used when the reference assigns ``.noise = 0.1`` /
    ``.lengthscale = l0`` (fs_mol/utils/gp_utils.py:17, fs_mol/models/adaptive_dkt.py:101)."""
    y = torch.as_tensor(y, dtype=DT)
    return y + torch.log(-torch.expm1(-y))


def transform_phi(phi: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """phi = (raw_noise, raw_outputscale, raw_lengthscale[1 or d]) in ``gp_params()`` order
    (fs_mol/models/adaptive_dkt.py:81-86; shapes SURVEY App. A2)."""
    noise = softplus(phi[0]) + NOISE_LOWER_BOUND
    outputscale = softplus(phi[1])
    lengthscale = softplus(phi[2:])
    return noise, outputscale, lengthscale


# --------------------------------------------------------------------------------------
# priors (fs_mol/models/adaptive_dkt.py:94-100, 112-121)
# --------------------------------------------------------------------------------------
@dataclass
class Priors:
    """LogNormal prior hyper-parameters.  ``ls_scale <= 0`` disables the lengthscale prior
    (``use_lengthscale_prior=False``)."""

    noise_loc: float
    noise_scale: float
    ls_loc: float = 0.0
    ls_scale: float = -1.0

    def as_array(self) -> np.ndarray:
        return np.array([self.noise_loc, self.noise_scale, self.ls_loc, self.ls_scale], dtype=np.float64)


def lognormal_log_prob(x: torch.Tensor, loc: float, scale: float) -> torch.Tensor:
    """gpytorch.priors.LogNormalPrior.log_prob (TransformedDistribution(Normal, Exp));
    SURVEY App. A4.  Summed over elements (ARD)."""
    lx = torch.log(x)
    return (-lx - math.log(scale) - 0.5 * LOG_2PI - (lx - loc) ** 2 / (2.0 * scale * scale)).sum()


def noise_prior_params(use_numeric_labels: bool) -> Tuple[float, float]:
    """fs_mol/models/adaptive_dkt.py:112-119: LogNormal(loc=log(mode)+scale^2, scale=0.25)."""
    scale = 0.25
    mode = 0.01 if use_numeric_labels else 0.1
    return math.log(mode) + scale ** 2, scale


# --------------------------------------------------------------------------------------
# median-heuristic lengthscale init (fs_mol/models/adaptive_dkt.py:128-131)
# --------------------------------------------------------------------------------------
def median_lengthscale_init(Z: torch.Tensor) -> torch.Tensor:
    d2 = torch.cdist(Z, Z) ** 2
    d2 = torch.triu(d2, diagonal=1)
    return torch.sqrt(0.5 * torch.median(d2[d2 > 0.0]))  # torch.median = LOWER median


def init_phi(Z_s: torch.Tensor, use_numeric_labels: bool = False, use_lengthscale_prior: bool = True,
             ard: bool = False) -> Tuple[torch.Tensor, Priors]:
    """Fresh per-task GP parameters and priors exactly as ``reinit_gp_params`` builds them
    (fs_mol/models/adaptive_dkt.py:88-126; fs_mol/utils/gp_utils.py:16-17):
    noise = 0.1 (cls) / 0.01 (reg), raw_outputscale = 0, lengthscale = median heuristic."""
    Z_s = Z_s.detach().to(DT)
    l0 = median_lengthscale_init(Z_s)
    n_loc, n_scale = noise_prior_params(use_numeric_labels)
    pri = Priors(n_loc, n_scale)
    if use_lengthscale_prior:
        pri.ls_scale = 0.25
        pri.ls_loc = math.log(l0.item()) + 0.25 ** 2
    noise0 = 0.01 if use_numeric_labels else 0.1
    raw_noise = inv_softplus(noise0 - NOISE_LOWER_BOUND)
    raw_ls = inv_softplus(l0)
    n_ls = Z_s.shape[1] if ard else 1
    phi = torch.cat([raw_noise.reshape(1), torch.zeros(1, dtype=DT), raw_ls.reshape(1).repeat(n_ls)])
    return phi, pri


# --------------------------------------------------------------------------------------
# kernels (fs_mol/utils/gp_utils.py:26-30 -> gpytorch ScaleKernel(RBF|Matern nu=2.5); App. A3)
# --------------------------------------------------------------------------------------
def scaled_sqdist(Z1: torch.Tensor, Z2: torch.Tensor, lengthscale: torch.Tensor) -> torch.Tensor:
    a = Z1 / lengthscale
    b = Z2 / lengthscale
    diff = a.unsqueeze(1) - b.unsqueeze(0)
    return (diff * diff).sum(-1)


def kernel_matrix(Z1: torch.Tensor, Z2: torch.Tensor, outputscale: torch.Tensor, lengthscale: torch.Tensor,
                  kind: int) -> torch.Tensor:
    u = scaled_sqdist(Z1, Z2, lengthscale)
    if kind == KERNEL_RBF:
        k = torch.exp(-0.5 * u)
    elif kind == KERNEL_MATERN52:
        # sqrt with a subgradient-safe floor: value identical, derivative wrt u finite at 0
        r = torch.sqrt(u.clamp_min(1e-30))
        k = (1.0 + math.sqrt(5.0) * r + (5.0 / 3.0) * u) * torch.exp(-math.sqrt(5.0) * r)
    else:
        raise ValueError(kind)
    return outputscale * k


# --------------------------------------------------------------------------------------
# f_inner: -ExactMarginalLogLikelihood (fs_mol/models/adaptive_dkt.py:173-176; App. A4/A5)
# --------------------------------------------------------------------------------------
def mvn_log_prob(y: torch.Tensor, mean: torch.Tensor, cov: torch.Tensor) -> torch.Tensor:
    L = torch.linalg.cholesky(cov)
    r = (y - mean).unsqueeze(-1)
    alpha = torch.cholesky_solve(r, L)
    n = y.shape[0]
    return -0.5 * (r * alpha).sum() - torch.log(torch.diagonal(L)).sum() - 0.5 * n * LOG_2PI


def f_inner(Z_s: torch.Tensor, y_s: torch.Tensor, phi: torch.Tensor, pri: Priors, kind: int) -> torch.Tensor:
    """-[log N(y;0,K+s2 I) + log p(noise) + log p(lengthscale)] / N  (priors added BEFORE /N)."""
    noise, os_, ls = transform_phi(phi)
    n = Z_s.shape[0]
    A = kernel_matrix(Z_s, Z_s, os_, ls, kind) + noise * torch.eye(n, dtype=Z_s.dtype)
    res = mvn_log_prob(y_s, torch.zeros_like(y_s), A)
    res = res + lognormal_log_prob(noise, pri.noise_loc, pri.noise_scale)
    if pri.ls_scale > 0:
        res = res + lognormal_log_prob(ls, pri.ls_loc, pri.ls_scale)
    return -res / n


# --------------------------------------------------------------------------------------
# exact prediction + f_outer (fs_mol/models/adaptive_dkt.py:183-191, 198-203; App. A6)
# --------------------------------------------------------------------------------------
def predict(Z_s, y_s, Z_q, phi, kind) -> Tuple[torch.Tensor, torch.Tensor]:
    """Posterior N(mu_q, Sigma_q) *with* likelihood noise added (``gp_likelihood(gp_model(x_q))``)."""
    noise, os_, ls = transform_phi(phi)
    n, m = Z_s.shape[0], Z_q.shape[0]
    A = kernel_matrix(Z_s, Z_s, os_, ls, kind) + noise * torch.eye(n, dtype=Z_s.dtype)
    Kqs = kernel_matrix(Z_q, Z_s, os_, ls, kind)
    Kqq = kernel_matrix(Z_q, Z_q, os_, ls, kind)
    L = torch.linalg.cholesky(A)
    alpha = torch.cholesky_solve(y_s.unsqueeze(-1), L).squeeze(-1)
    mean = Kqs @ alpha
    V = torch.cholesky_solve(Kqs.T, L)
    cov = Kqq - Kqs @ V + noise * torch.eye(m, dtype=Z_s.dtype)
    return mean, cov


def f_outer(Z_s, y_s, Z_q, y_q, phi, kind) -> torch.Tensor:
    """-log N(y_q; mu_q, Sigma_q + noise I): the JOINT density, un-normalised by N_q
    (fs_mol/models/adaptive_dkt.py:189)."""
    mean, cov = predict(Z_s, y_s, Z_q, phi, kind)
    return -mvn_log_prob(y_q, mean, cov)


# --------------------------------------------------------------------------------------
# inner fit: fit_gpytorch_scipy (fs_mol/utils/adaptive_dkt_utils.py:91; App. A7)
# --------------------------------------------------------------------------------------
def fit_phi(Z_s, y_s, phi0, pri, kind, maxiter: int = 15000, dtype=DT):
    """SciPy L-BFGS-B on -mll with SciPy defaults (float64 host vector, as BoTorch does)."""
    from scipy.optimize import minimize

    Z = Z_s.detach().to(dtype)
    y = y_s.detach().to(dtype)

    def fun(x):
        p = torch.tensor(x, dtype=dtype, requires_grad=True)
        loss = f_inner(Z, y, p, pri, kind)
        (g,) = torch.autograd.grad(loss, p)
        return float(loss.item()), g.double().numpy().astype(np.float64)

    res = minimize(fun, phi0.detach().double().numpy(), jac=True, method="L-BFGS-B",
                   options={"maxiter": maxiter, "maxfun": 15000})
    return torch.tensor(res.x, dtype=DT), res


# --------------------------------------------------------------------------------------
# everything the HIP path must reproduce at fixed (Z, y, phi), via autograd
# --------------------------------------------------------------------------------------
def full_reference_quantities(Z_s, y_s, Z_q, y_q, phi, pri: Priors, kind: int) -> dict:
    """Returns f_in, grad_phi f_in, H, f_out, grad_phi f_out, v = H^-1 grad_phi f_out, direct
    d f_out/dZ, the mixed VJP d(v^T grad_phi f_in)/dZ_s and the total IFT dL/dZ
    (= cauchy_hypergradient at the feature-matrix level, fs_mol/utils/cauchy_hypergradient.py:43-161),
    plus predictive mean / covariance diagonal."""
    Z_s = Z_s.detach().to(DT).requires_grad_(True)
    Z_q = Z_q.detach().to(DT).requires_grad_(True)
    y_s = y_s.detach().to(DT)
    y_q = y_q.detach().to(DT)
    phi = phi.detach().to(DT).requires_grad_(True)

    fin = f_inner(Z_s, y_s, phi, pri, kind)
    (g_in,) = torch.autograd.grad(fin, phi, create_graph=True)
    H = torch.stack([torch.autograd.grad(g_in[i], phi, retain_graph=True)[0] for i in range(phi.numel())])
    (dfin_dZs,) = torch.autograd.grad(fin, Z_s, retain_graph=True)

    fout = f_outer(Z_s, y_s, Z_q, y_q, phi, kind)
    g_out, dZs_direct, dZq_direct = torch.autograd.grad(fout, (phi, Z_s, Z_q))

    v = torch.linalg.solve(H.detach(), g_out)
    (mixed_Zs,) = torch.autograd.grad((g_in * v.detach()).sum(), Z_s)

    mean, cov = predict(Z_s.detach(), y_s, Z_q.detach(), phi.detach(), kind)
    return {
        "l0": median_lengthscale_init(Z_s.detach()).item(),
        "f_in": fin.item(),
        "g_in": g_in.detach().numpy().copy(),
        "dfin_dZs": dfin_dZs.numpy().copy(),
        "H": H.detach().numpy().copy(),
        "f_out": fout.item(),
        "g_out": g_out.numpy().copy(),
        "v": v.numpy().copy(),
        "dZs_direct": dZs_direct.numpy().copy(),
        "dZq_direct": dZq_direct.numpy().copy(),
        "mixed_Zs": mixed_Zs.numpy().copy(),
        "dZs_total": (dZs_direct - mixed_Zs).numpy().copy(),
        "dZq_total": dZq_direct.numpy().copy(),
        "pred_mean": mean.numpy().copy(),
        "pred_var": torch.diagonal(cov).numpy().copy(),
        "pred_cov": cov.numpy().copy(),
    }
