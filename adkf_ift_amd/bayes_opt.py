"""The Bayesian-optimisation caller of the same GP op: ``create_gp`` + the GP-EI loop of
bayes_opt/bo_utils.py:342-455, on the HIP library.  The GP is fitted on the queried points, Expected Improvement is
evaluated for EVERY candidate with one ``adkf_predict`` call (the reference loops over candidates one by one).

Only the Matern-5/2 branch exists here (the Tanimoto kernel of the reference's fingerprint baseline is not a
distance-based kernel and is out of the library's scope).
"""
from __future__ import annotations

import math
from typing import List, Optional, Tuple

import numpy as np
import torch

from . import gp_ops
from .models import ExactGPLayer, ExactMarginalLogLikelihood, GaussianLikelihood, fit_gpytorch_scipy


def compute_median_lengthscale_init(gp_input: torch.Tensor) -> torch.Tensor:
    """bo_utils.py:458-461 (same median heuristic as the model's)."""
    b = gp_ops.GPBatch(gp_input.detach()[None].float().contiguous(), torch.zeros(1, gp_input.shape[0], device=gp_input.device),
                       torch.zeros(1, 4, device=gp_input.device), gp_ops.KERNEL_RBF)
    return gp_ops.median_lengthscale(b)[0]


def create_gp(train_x: torch.Tensor, train_y: torch.Tensor, kernel_type: str, device, noise_init: float, noise_prior: bool):
    """bo_utils.py:423-455: (likelihood, model, mll) with noise initialised at ``noise_init`` (optionally under a
    LogNormal prior with that mode) and a Matern-5/2 kernel whose lengthscale starts at, and has its prior mode at, the
    median heuristic of ``train_x``."""
    if kernel_type != "matern":
        raise ValueError(f"kernel_type {kernel_type!r}: only 'matern' is available on the HIP path")
    scale = 0.25
    prior = (math.log(noise_init) + scale ** 2, scale) if noise_prior else None
    likelihood = GaussianLikelihood(noise_prior=prior).to(device)
    model = ExactGPLayer(train_x, train_y, likelihood, "matern").to(device)
    likelihood.noise = noise_init
    l0 = compute_median_lengthscale_init(train_x)
    bk = model.covar_module.base_kernel
    bk.register_prior("lengthscale_prior", (torch.log(l0).item() + scale ** 2, scale))
    bk.lengthscale = torch.ones_like(bk.lengthscale) * l0
    mll = ExactMarginalLogLikelihood(likelihood, model).to(device)
    return likelihood, model, mll


@torch.no_grad()
def latent_posterior(model: ExactGPLayer, mll: ExactMarginalLogLikelihood, X: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
    """Mean and variance of the LATENT function at X (``model.posterior(X)`` without observation noise, which is what
    BoTorch's analytic acquisition functions read)."""
    Z = model.train_inputs[0].detach().float().contiguous()
    y = model.train_targets.detach().float().contiguous()
    b = gp_ops.GPBatch(Z[None], y[None], mll.priors_row(Z.device), model.kernel_id, Z_q=X.detach().float().contiguous()[None])
    phi = torch.cat([p.detach().reshape(-1) for p in mll.raw_params()])[None]
    mean, var, _, info = gp_ops.predict(b, phi, want_var=True)
    gp_ops.check_info(info, "BO posterior")
    noise = model.likelihood.noise.detach().reshape(())
    return mean[0], (var[0] - noise).clamp_min(1e-12)


def expected_improvement(mean: torch.Tensor, var: torch.Tensor, best_f: float, maximize: bool = False) -> torch.Tensor:
    """botorch.acquisition.analytic.ExpectedImprovement: sigma * (u Phi(u) + phi(u)), u = +-(mean - best_f) / sigma."""
    sigma = var.sqrt()
    u = (mean - best_f) / sigma
    if not maximize:
        u = -u
    normal = torch.distributions.Normal(torch.zeros_like(u), torch.ones_like(u))
    return sigma * (u * normal.cdf(u) + torch.exp(normal.log_prob(u)))


def run_gp_ei_bo(x_all: torch.Tensor, y_all: torch.Tensor, num_init_points: int, query_batch_size: int, num_bo_iters: int,
                 kernel_type: str, device, init_from: int, noise_init: float, noise_prior: bool,
                 rng: Optional[np.random.Generator] = None) -> List[int]:
    """bo_utils.py:342-397 (minimisation; points sorted by ascending y).  Returns the BO record: the best initial index
    followed by the queried indices in the order the reference appends them."""
    rng = rng or np.random.default_rng()
    n = x_all.shape[0]
    y_all = (y_all - y_all.mean()) / y_all.std()
    queried = rng.choice(np.arange(init_from, n), size=num_init_points, replace=False).tolist()
    record = [min(queried)]
    for _ in range(num_bo_iters):
        xq, yq = x_all[queried], y_all[queried]
        best = yq.min().item()
        likelihood, model, mll = create_gp(xq, yq, kernel_type, device, noise_init, noise_prior)
        fit_gpytorch_scipy(mll)
        mean, var = latent_posterior(model, mll, x_all)
        acq = expected_improvement(mean, var, best, maximize=False).cpu()
        acq[queried] = -float("inf")
        nonzero = int((acq > 0).sum())
        free = lambda taken: [i for i in range(n) if i not in taken]
        if nonzero == 0:
            pick = rng.choice(free(queried), size=query_batch_size, replace=False).tolist()
        elif nonzero < query_batch_size:
            pick = torch.topk(acq, query_batch_size).indices[:nonzero].tolist()
            pick += rng.choice(free(queried + pick), size=query_batch_size - nonzero, replace=False).tolist()
        else:
            pick = torch.topk(acq, query_batch_size).indices.tolist()
        queried = list(set(queried + pick))
        record.extend(pick[::-1])
    return record
