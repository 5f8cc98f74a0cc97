"""Deep-kernel models with the reference's operator surface, GP tail on the HIP library.

Mirrors (names, arguments, modes, return conventions):
  * ``ADKTModel``        fs_mol/models/adaptive_dkt.py:36-209
  * ``DKLModel``         fs_mol/models/dkl.py:37-161 (surface only: ``forward(batch, train)`` / ``compute_loss``)
  * ``ExactGPLayer``     fs_mol/utils/gp_utils.py:7-49 (parameter container: the arithmetic is in csrc/)
  * ``fit_gpytorch_scipy`` botorch.optim.fit.fit_gpytorch_scipy as called at fs_mol/utils/adaptive_dkt_utils.py:91

The GP is no longer a GPyTorch module, so the three raw parameters keep their GPyTorch names
(``gp_likelihood.noise_covar.raw_noise``, ``gp_model.covar_module.raw_outputscale``,
``gp_model.covar_module.base_kernel.raw_lengthscale``; shapes [1], [], [1,1]) and ``gp_params()`` returns them in
that order, which is what checkpoints and the trainers of the reference rely on.  ARD (``use_ard``: ``raw_lengthscale``
of shape [1, d]) runs on the HVP + conjugate-gradient path of the library (csrc/ard.h).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Any, List, Optional, Tuple

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import gp_ops
from .stateless import functional_call

FINGERPRINT_DIM = 2048
PHYS_CHEM_DESCRIPTORS_DIM = 42
NOISE_LOWER_BOUND = 1e-4


def _inv_softplus(y: torch.Tensor) -> torch.Tensor:
    return y + torch.log(-torch.expm1(-y))


# ----------------------------------------------------------------------------------------------------------
# parameter containers (GPyTorch names)
# ----------------------------------------------------------------------------------------------------------
class _NoiseCovar(nn.Module):
    def __init__(self):
        super().__init__()
        self.raw_noise = nn.Parameter(torch.zeros(1))

    @property
    def noise(self):
        return F.softplus(self.raw_noise) + NOISE_LOWER_BOUND

    @noise.setter
    def noise(self, value):
        with torch.no_grad():
            v = torch.as_tensor(value, dtype=self.raw_noise.dtype, device=self.raw_noise.device)
            self.raw_noise.copy_(_inv_softplus(v - NOISE_LOWER_BOUND).reshape(1))


class GaussianLikelihood(nn.Module):
    """gpytorch.likelihoods.GaussianLikelihood(noise_prior=LogNormalPrior(loc, scale)) as a parameter holder."""

    def __init__(self, noise_prior: Optional[Tuple[float, float]] = None):
        super().__init__()
        self.noise_covar = _NoiseCovar()
        self.noise_prior = noise_prior  # (loc, scale) or None

    @property
    def noise(self):
        return self.noise_covar.noise

    @noise.setter
    def noise(self, v):
        self.noise_covar.noise = v


class _BaseKernel(nn.Module):
    def __init__(self, ard_num_dims=None):
        super().__init__()
        self.ard_num_dims = ard_num_dims   # gpytorch: lengthscale [1, d] instead of [1, 1]
        self.raw_lengthscale = nn.Parameter(torch.zeros(1, ard_num_dims or 1))
        self.lengthscale_prior: Optional[Tuple[float, float]] = None

    @property
    def lengthscale(self):
        return F.softplus(self.raw_lengthscale)

    @lengthscale.setter
    def lengthscale(self, value):
        with torch.no_grad():
            v = torch.as_tensor(value, dtype=self.raw_lengthscale.dtype, device=self.raw_lengthscale.device)
            self.raw_lengthscale.copy_(_inv_softplus(v).reshape(1, -1).expand_as(self.raw_lengthscale))

    def register_prior(self, name, prior: Tuple[float, float], *unused):
        self.lengthscale_prior = prior


class _ScaleKernel(nn.Module):
    def __init__(self, ard_num_dims=None):
        super().__init__()
        self.raw_outputscale = nn.Parameter(torch.zeros(()))
        self.base_kernel = _BaseKernel(ard_num_dims)

    @property
    def outputscale(self):
        return F.softplus(self.raw_outputscale)


class ExactGPLayer(nn.Module):
    """ZeroMean + ScaleKernel(RBF | Matern-5/2) exact GP: holds parameters and the current train data."""

    def __init__(self, train_x, train_y, likelihood: GaussianLikelihood, kernel: str, ard_num_dims=None,
                 use_numeric_labels: bool = False):
        super().__init__()
        self.kernel_id = gp_ops.kernel_id(kernel)
        self.ard = ard_num_dims is not None
        likelihood.noise_covar.raw_noise.requires_grad = True
        likelihood.noise = 0.01 if use_numeric_labels else 0.1      # fs_mol/utils/gp_utils.py:17
        self.likelihood = likelihood
        self.covar_module = _ScaleKernel(ard_num_dims)
        self.train_inputs = (train_x,)
        self.train_targets = train_y

    def set_train_data(self, inputs=None, targets=None, strict=False):
        if inputs is not None:
            self.train_inputs = (inputs,)
        if targets is not None:
            self.train_targets = targets


class ExactMarginalLogLikelihood(nn.Module):
    """Handle passed to ``fit_gpytorch_scipy``; calling it evaluates the MLL of the stored model on the GPU."""

    def __init__(self, likelihood: GaussianLikelihood, model: ExactGPLayer):
        super().__init__()
        self.likelihood = likelihood
        self.model = model

    def priors_row(self, device) -> torch.Tensor:
        n = self.likelihood.noise_prior or (0.0, -1.0)
        l = self.model.covar_module.base_kernel.lengthscale_prior or (0.0, -1.0)
        return torch.tensor([[n[0], n[1], l[0], l[1]]], dtype=torch.float32, device=device)

    def raw_params(self) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
        return (self.likelihood.noise_covar.raw_noise, self.model.covar_module.raw_outputscale,
                self.model.covar_module.base_kernel.raw_lengthscale)

    def forward(self, function_dist: "GPTrainHandle", target: torch.Tensor) -> torch.Tensor:
        phi = torch.cat([p.reshape(-1) for p in self.raw_params()])
        return _MLLFunction.apply(function_dist.Z, target, phi, self.priors_row(function_dist.Z.device),
                                  self.model.kernel_id, self.model.ard)


@dataclass
class GPTrainHandle:
    """What ``gp_model(train_x)`` returns in training mode: the features the prior is evaluated on."""

    Z: torch.Tensor


class GPPosterior:
    """What eval-mode ``forward`` returns (the reference returns a gpytorch MultivariateNormal): ``.mean``,
    ``.variance``, ``.covariance_matrix`` (both with the likelihood noise added) and ``.log_prob``."""

    def __init__(self, mean, variance, covariance_matrix):
        self.mean, self.variance, self.covariance_matrix = mean, variance, covariance_matrix

    @property
    def stddev(self):
        return self.variance.sqrt()

    def log_prob(self, y: torch.Tensor) -> torch.Tensor:
        L = torch.linalg.cholesky(self.covariance_matrix)
        r = (y - self.mean).unsqueeze(-1)
        a = torch.cholesky_solve(r, L)
        n = y.shape[-1]
        return -0.5 * (r * a).sum() - torch.log(torch.diagonal(L)).sum() - 0.5 * n * math.log(2 * math.pi)


# ----------------------------------------------------------------------------------------------------------
# autograd bridges (first order; second-order quantities come from adkf_ift_hypergrad, not from autograd)
# ----------------------------------------------------------------------------------------------------------
class _MLLFunction(torch.autograd.Function):
    """+MLL (so that callers negate it like the reference does) of one task; differentiable in Z and phi."""

    @staticmethod
    def forward(ctx, Z, y, phi, priors, kernel, ard=False):
        b = gp_ops.GPBatch(Z.detach()[None], y.detach()[None].float(), priors, kernel, ard=ard)
        f, g, dZ, info = gp_ops.mll_value_grad(b, phi.detach()[None], want_dZ=True)
        gp_ops.check_info(info, "marginal log likelihood")
        ctx.save_for_backward(g[0], dZ[0])
        return -f[0]

    @staticmethod
    def backward(ctx, grad_out):
        g, dZ = ctx.saved_tensors
        return -grad_out * dZ, None, -grad_out * g, None, None, None


class _OuterNLLFunction(torch.autograd.Function):
    """f_outer = -log N(y_q; mu_q, Sigma_q + noise I) of one task; differentiable in Z_s, Z_q and phi."""

    @staticmethod
    def forward(ctx, Z_s, y_s, Z_q, y_q, phi, priors, kernel, ard=False):
        b = gp_ops.GPBatch(Z_s.detach()[None], y_s.detach()[None].float(), priors, kernel, Z_q=Z_q.detach()[None],
                           y_q=y_q.detach()[None].float(), ard=ard)
        f, g, dZs, dZq, info = gp_ops.outer_nll_value_grad(b, phi.detach()[None])
        gp_ops.check_info(info, "predictive log likelihood")
        ctx.save_for_backward(g[0], dZs[0], dZq[0])
        return f[0]

    @staticmethod
    def backward(ctx, grad_out):
        g, dZs, dZq = ctx.saved_tensors
        return grad_out * dZs, None, grad_out * dZq, None, grad_out * g, None, None, None


def fit_gpytorch_scipy(mll: ExactMarginalLogLikelihood, max_evals: int = 200, gtol: float = 1e-5, ftol: float = gp_ops.FTOL_DEFAULT):
    """Drop-in for ``botorch.optim.fit.fit_gpytorch_scipy(model.mll)``: minimises -mll over the raw GP parameters on the
    GPU (in-kernel BFGS for the three-parameter kernel, device L-BFGS for ARD) and writes the optimum back into the
    module.  Returns ``(mll, info_dict)`` like BoTorch."""
    model = mll.model
    Z = model.train_inputs[0].detach()
    y = model.train_targets.detach().float()
    b = gp_ops.GPBatch(Z[None], y[None], mll.priors_row(Z.device), model.kernel_id, ard=model.ard)
    phi0 = torch.cat([p.detach().reshape(-1) for p in mll.raw_params()])[None]
    phi, f, gnorm, nev, info = gp_ops.fit(b, phi0, max_evals, gtol, ftol)
    gp_ops.check_info(info, "fit_gpytorch_scipy")
    with torch.no_grad():
        off = 0
        for p in mll.raw_params():
            p.copy_(phi[0, off:off + p.numel()].reshape(p.shape))
            off += p.numel()
    return mll, {"fopt": f[0].item(), "max_abs_grad": gnorm[0].item(), "nfev": int(nev[0].item())}


# ----------------------------------------------------------------------------------------------------------
# models
# ----------------------------------------------------------------------------------------------------------
@dataclass(frozen=True)
class ADKTModelConfig:
    """fs_mol/models/adaptive_dkt.py:26-33 plus the GP fields the model reads from the trainer config
    (fs_mol/utils/adaptive_dkt_utils.py:63-67)."""

    graph_feature_extractor_config: Any = None
    used_features: str = "gnn+ecfp+fc"
    use_ard: bool = False
    gp_kernel: str = "matern"
    use_lengthscale_prior: bool = True
    use_numeric_labels: bool = False
    ignore_grad_correction: bool = False
    fc_hidden_dim: int = 2048       # reference: fixed 2048 (adaptive_dkt.py:61-65); configurable for tests
    fc_out_dim: int = 2048


class _DeepKernelBase(nn.Module):
    """Feature head shared by ADKTModel and DKLModel (adaptive_dkt.py:41-65,141-164 / dkl.py:42-66,108-131)."""

    def _build_features(self, config):
        if config.used_features.startswith("gnn"):
            from .gnn import GraphFeatureExtractor, GraphFeatureExtractorConfig
            gcfg = config.graph_feature_extractor_config or GraphFeatureExtractorConfig()
            self.graph_feature_extractor = GraphFeatureExtractor(gcfg)
            gnn_out = gcfg.readout_config.output_dim
        else:
            gnn_out = 0
        self.use_fc = config.used_features.endswith("+fc")
        self.fc_out_dim = config.fc_out_dim
        if self.use_fc:
            fc_in = 0
            if "gnn" in config.used_features:
                fc_in += gnn_out
            if "ecfp" in config.used_features:
                fc_in += FINGERPRINT_DIM
            if "pc-descs" in config.used_features:
                fc_in += PHYS_CHEM_DESCRIPTORS_DIM
            self.fc = nn.Sequential(nn.Linear(fc_in, config.fc_hidden_dim), nn.ReLU(),
                                    nn.Linear(config.fc_hidden_dim, config.fc_out_dim))
        self.normalizing_features = config.gp_kernel == "cossim"

    def _features(self, part) -> torch.Tensor:
        feats: List[torch.Tensor] = []
        uf = self.config.used_features
        if "gnn" in uf:
            feats.append(self.graph_feature_extractor(part))
        if "ecfp" in uf:
            feats.append(part.fingerprints.float())
        if "pc-descs" in uf:
            feats.append(part.descriptors)
        flat = torch.cat(feats, dim=1)
        if self.use_fc:
            flat = self.fc(flat)
        if self.normalizing_features:
            flat = F.normalize(flat, p=2, dim=1)
        return flat

    def _labels(self, batch):
        if self.config.use_numeric_labels:
            return batch.support_numeric_labels.float(), batch.query_numeric_labels.float()
        cv = lambda l: (l.float() - 0.5) * 2.0  # True -> 1.0; False -> -1.0
        return cv(batch.support_labels), cv(batch.query_labels)

    @property
    def device(self) -> torch.device:
        return next(self.parameters()).device

    def _posterior(self, Z_s, y_s, Z_q) -> GPPosterior:
        b = gp_ops.GPBatch(Z_s.detach()[None], y_s[None], self.mll.priors_row(Z_s.device), self.gp_model.kernel_id,
                           Z_q=Z_q.detach()[None], ard=self.gp_model.ard)
        phi = torch.cat([p.detach().reshape(-1) for p in self.mll.raw_params()])[None]
        mean, var, cov, info = gp_ops.predict(b, phi, want_cov=True)
        gp_ops.check_info(info, "GP prediction")
        return GPPosterior(mean[0], var[0], cov[0])


class ADKTModel(_DeepKernelBase):
    def __init__(self, config: ADKTModelConfig):
        super().__init__()
        self.config = config
        self._build_features(config)
        self.__create_tail_GP(kernel_type=config.gp_kernel)

    def feature_extractor_params(self):
        return [p for n, p in self.named_parameters() if not n.startswith("gp_")]

    def gp_params(self):
        return [p for n, p in self.named_parameters() if n.startswith("gp_")]

    def __create_tail_GP(self, kernel_type):
        dev = next(self.parameters(), torch.zeros(1)).device
        scale = 0.25
        mode = 0.01 if self.config.use_numeric_labels else 0.1
        noise_prior = (math.log(mode) + scale ** 2, scale)          # LogNormal with mode = `mode` (adaptive_dkt.py:112-119)
        ard = self.fc_out_dim if self.config.use_ard else None
        self.gp_likelihood = GaussianLikelihood(noise_prior=noise_prior).to(dev)
        self.gp_model = ExactGPLayer(torch.ones(64, self.fc_out_dim), torch.ones(64), self.gp_likelihood, kernel_type,
                                     ard_num_dims=ard, use_numeric_labels=self.config.use_numeric_labels).to(dev)
        self.mll = ExactMarginalLogLikelihood(self.gp_likelihood, self.gp_model).to(dev)

    def compute_median_lengthscale_init(self, gp_input: torch.Tensor) -> torch.Tensor:
        b = gp_ops.GPBatch(gp_input.detach()[None], torch.zeros(1, gp_input.shape[0], device=gp_input.device),
                           torch.zeros(1, 4, device=gp_input.device), gp_ops.KERNEL_RBF)
        return gp_ops.median_lengthscale(b)[0]

    def reinit_gp_params(self, gp_input, use_lengthscale_prior=False):
        self.__create_tail_GP(kernel_type=self.config.gp_kernel)
        if self.config.gp_kernel in ("matern", "rbf", "RBF"):
            l0 = self.compute_median_lengthscale_init(gp_input)
            if use_lengthscale_prior:
                scale = 0.25
                loc = torch.log(l0).item() + scale ** 2            # mode = l0 (adaptive_dkt.py:94-96)
                self.gp_model.covar_module.base_kernel.register_prior("lengthscale_prior", (loc, scale))
            bk = self.gp_model.covar_module.base_kernel
            bk.lengthscale = torch.ones_like(bk.lengthscale) * l0

    def forward(self, input_batch, train_loss: Optional[bool], predictive_val_loss: bool = False,
                is_functional_call: bool = False):
        return self._tail(*self._task_tensors(input_batch), train_loss, predictive_val_loss, is_functional_call)

    def _task_tensors(self, batch):
        """(Z_s, y_s, Z_q, y_q) of one task."""
        y_s, y_q = self._labels(batch)
        return self._features(batch.support_features), y_s, self._features(batch.query_features), y_q

    def _tail(self, Z_s, y_s, Z_q, y_q, train_loss: Optional[bool], predictive_val_loss: bool, is_functional_call: bool):
        """The four modes of ``ADKTModel.forward`` (adaptive_dkt.py:166-205) on given features."""
        if self.training:
            assert train_loss is not None
            if train_loss:
                if is_functional_call:   # f_inner
                    self.gp_model.set_train_data(inputs=Z_s, targets=y_s, strict=False)
                    logits = -self.mll(GPTrainHandle(Z_s), self.gp_model.train_targets)
                else:
                    self.reinit_gp_params(Z_s.detach(), self.config.use_lengthscale_prior)
                    self.gp_model.set_train_data(inputs=Z_s.detach(), targets=y_s.detach(), strict=False)
                    logits = None
            else:
                assert is_functional_call
                if predictive_val_loss:  # f_outer: joint predictive NLL, un-normalised (adaptive_dkt.py:183-191)
                    self.gp_model.set_train_data(inputs=Z_s, targets=y_s, strict=False)
                    phi = torch.cat([p.reshape(-1) for p in self.mll.raw_params()])
                    logits = _OuterNLLFunction.apply(Z_s, y_s, Z_q, y_q, phi, self.mll.priors_row(Z_s.device),
                                                     self.gp_model.kernel_id, self.gp_model.ard)
                else:
                    self.gp_model.set_train_data(inputs=Z_q, targets=y_q, strict=False)
                    logits = -self.mll(GPTrainHandle(Z_q), self.gp_model.train_targets)
        else:
            assert train_loss is None
            self.gp_model.set_train_data(inputs=Z_s, targets=y_s, strict=False)
            with torch.no_grad():
                logits = self._posterior(Z_s, y_s, Z_q)
        return logits

    # ---- fused IFT path ----------------------------------------------------------------------------------
    def task_losses(self, batch) -> Tuple["GPTaskLoss", "GPTaskLoss"]:
        """(f_outer, f_inner) for ``cauchy_hypergradient``: callables ``f(params_outer, params_inner)`` exactly like
        the closures at fs_mol/utils/adaptive_dkt_utils.py:383-395, tagged so that the fused HIP path is taken."""
        task = _FusedTask(self, batch)
        return GPTaskLoss(task, outer=True), GPTaskLoss(task, outer=False)


class _FusedTask:
    def __init__(self, model: ADKTModel, batch):
        self.model, self.batch = model, batch
        self.fe_names = [n for n, _ in model.named_parameters() if not n.startswith("gp_")]
        self.gp_names = [n for n, _ in model.named_parameters() if n.startswith("gp_")]

    def param_dict(self, params_outer, params_inner):
        d = {n: p for n, p in zip(self.fe_names, params_outer)}
        d.update({n: p for n, p in zip(self.gp_names, params_inner)})
        return d

    def fused_hypergradient(self, params_outer, params_inner, ignore_grad_correction, sanity_checks, ignore_direct_grad):
        m = self.model
        fe = {n: p for n, p in zip(self.fe_names, params_outer)}

        Z_s, y_s, Z_q, y_q = functional_call(_Features(m), {"m." + n: p for n, p in fe.items()}, (self.batch,))
        phi = torch.cat([p.detach().reshape(-1) for p in params_inner])[None]
        b = gp_ops.GPBatch(Z_s.detach()[None], y_s[None], m.mll.priors_row(Z_s.device), m.gp_model.kernel_id,
                           Z_q=Z_q.detach()[None], y_q=y_q[None], ard=m.gp_model.ard)
        out = gp_ops.ift_hypergrad(b, phi, ignore_grad_correction, ignore_direct_grad)
        gp_ops.check_info(out["info"], "cauchy_hypergradient")   # ARD: also raises when CG met non-positive curvature
        if sanity_checks and not ignore_grad_correction and out["H"] is not None:
            logabsdet = torch.linalg.slogdet(out["H"][0].double()).logabsdet
            assert logabsdet.item() > -10.0
        outer = [p for p in params_outer]
        torch.autograd.backward([Z_s, Z_q], [out["dZ_s"][0].to(Z_s.dtype), out["dZ_q"][0].to(Z_q.dtype)], inputs=outer)
        for p in outer:
            if p.grad is None:
                p.grad = torch.zeros_like(p)
        off = 0
        for p in params_inner:
            p.grad = out["g_phi"][0, off:off + p.numel()].reshape(p.shape).to(p.dtype).clone()
            off += p.numel()
        return out["f_out"][0]


class _Features(nn.Module):
    def __init__(self, m):
        super().__init__()
        self.m = m

    def forward(self, batch, outer: Optional[bool] = None):
        t = self.m._task_tensors(batch)
        if outer is None:
            return t
        return self.m._tail(*t, train_loss=not outer, predictive_val_loss=outer, is_functional_call=True)


class GPTaskLoss:
    """f(params_outer, params_inner) -> scalar, evaluated through ``functional_call`` like the reference's
    closures; also usable with the generic autograd path for first-order quantities."""

    def __init__(self, task: _FusedTask, outer: bool):
        self.task, self.outer = task, outer

    def __call__(self, params_outer, params_inner):
        params = {"m." + n: p for n, p in self.task.param_dict(params_outer, params_inner).items()}
        return functional_call(_Features(self.task.model), params, (self.task.batch, self.outer))


@dataclass(frozen=True)
class DKLModelConfig(ADKTModelConfig):
    gp_kernel: str = "matern"
    use_lengthscale_prior: bool = False


class DKLModel(_DeepKernelBase):
    """fs_mol/models/dkl.py: same GP tail trained jointly with the features (no IFT)."""

    def __init__(self, config: DKLModelConfig):
        super().__init__()
        self.config = config
        self._build_features(config)
        self.gp_likelihood = GaussianLikelihood()                    # no noise prior (dkl.py:86)
        self.gp_model = ExactGPLayer(torch.ones(64, self.fc_out_dim), torch.ones(64), self.gp_likelihood, config.gp_kernel,
                                     ard_num_dims=self.fc_out_dim if config.use_ard else None,
                                     use_numeric_labels=config.use_numeric_labels)
        if config.use_lengthscale_prior:                              # LogNormal(0, 0.25), init at its mean (dkl.py:92-99)
            bk = self.gp_model.covar_module.base_kernel
            bk.register_prior("lengthscale_prior", (0.0, 0.25))
            bk.lengthscale = torch.ones_like(bk.lengthscale) * math.exp(0.0 + 0.25 ** 2 / 2)
        self.mll = ExactMarginalLogLikelihood(self.gp_likelihood, self.gp_model)

    def forward(self, input_batch, train: bool):
        Z_s = self._features(input_batch.support_features)
        Z_q = self._features(input_batch.query_features)
        y_s, _ = self._labels(input_batch)
        if self.training and train:
            self.gp_model.set_train_data(inputs=Z_s, targets=y_s, strict=False)
            return GPTrainHandle(Z_s)
        assert not self.training and not train
        self.gp_model.set_train_data(inputs=Z_s.detach(), targets=y_s, strict=False)
        with torch.no_grad():
            return self._posterior(Z_s, y_s, Z_q)

    def compute_loss(self, logits: GPTrainHandle) -> torch.Tensor:
        assert self.training
        return -self.mll(logits, self.gp_model.train_targets)


@dataclass(frozen=True)
class DKTModelConfig(DKLModelConfig):
    pass


class DKTModel(DKLModel):
    """fs_mol/models/dkt.py: Deep Kernel Transfer - GP hyper-parameters SHARED across tasks and trained with the
    features on the joint marginal likelihood of support and query points (dkt.py:146-151); at test time the GP is
    conditioned on the support set, optionally after re-fitting the hyper-parameters from their saved meta-learned
    values (``test_time_adaptation``, dkt.py:152-168)."""

    def __init__(self, config: DKTModelConfig):
        super().__init__(config)
        self.test_time_adaptation = False
        self.gp_model_params = None
        self.gp_likelihood_params = None

    def save_gp_params(self):
        from copy import deepcopy
        self.gp_model_params = deepcopy(self.gp_model.state_dict())
        self.gp_likelihood_params = deepcopy(self.gp_likelihood.state_dict())

    def load_gp_params(self):
        self.gp_model.load_state_dict(self.gp_model_params)
        self.gp_likelihood.load_state_dict(self.gp_likelihood_params)

    def forward(self, input_batch):
        Z_s = self._features(input_batch.support_features)
        Z_q = self._features(input_batch.query_features)
        y_s, y_q = self._labels(input_batch)
        if self.training:
            Z = torch.cat([Z_s, Z_q], dim=0)
            self.gp_model.set_train_data(inputs=Z, targets=torch.cat([y_s, y_q]), strict=False)
            return GPTrainHandle(Z)
        if self.test_time_adaptation:
            self.load_gp_params()
        self.gp_model.set_train_data(inputs=Z_s.detach(), targets=y_s, strict=False)
        if self.test_time_adaptation:
            fit_gpytorch_scipy(self.mll)
        with torch.no_grad():
            return self._posterior(Z_s, y_s, Z_q)


class ADKFModel(ADKTModel):
    """MoleculeNet variant (MoleculeNet/chem_lib/models/adkf_model.py:14-159): the same adaptive GP tail behind a
    pluggable molecule encoder ``mol_encoder(x, edge_index, edge_attr, batch) -> (features, _)`` (the reference plugs in a
    pre-trained GIN from torch_geometric; any module with that signature works).  Classification labels only,
    lengthscale prior always on, no ARD (adkf_model.py:55,72)."""

    def __init__(self, mol_encoder: nn.Module, emb_dim: int, gp_kernel: str = "matern"):
        nn.Module.__init__(self)
        self.mol_encoder = mol_encoder
        self.emb_dim = self.fc_out_dim = emb_dim
        self.gp_kernel = gp_kernel
        self.config = ADKTModelConfig(used_features="encoder", gp_kernel=gp_kernel, use_lengthscale_prior=True,
                                      use_numeric_labels=False, fc_out_dim=emb_dim)
        self._ADKTModel__create_tail_GP(kernel_type=gp_kernel)

    @staticmethod
    def _convert_bool_labels(labels):
        return (labels.float() - 0.5) * 2.0

    def _encode(self, data):
        return self.mol_encoder(data.x, data.edge_index, data.edge_attr, data.batch)[0]

    def _task_tensors(self, batch):
        """batch = (s_data, q_data, s_label), the arguments of ``forward`` (adkfift_trainer.py:177-199)."""
        s_data, q_data, s_label = batch
        return self._encode(s_data), self._convert_bool_labels(s_label), self._encode(q_data), self._convert_bool_labels(q_data.y)

    def forward(self, s_data, q_data, train_loss: Optional[bool], s_label=None, q_pred_adj=False,
                predictive_val_loss: bool = False, is_functional_call: bool = False):
        assert self.training and train_loss is not None      # adkf_model.py:104-105
        Z_s = self._encode(s_data)
        y_s = self._convert_bool_labels(s_label)
        Z_q = y_q = None
        if q_data is not None:
            Z_q, y_q = self._encode(q_data), self._convert_bool_labels(q_data.y)
        if not train_loss and not predictive_val_loss:
            raise NotImplementedError                          # adkf_model.py:127-131
        return self._tail(Z_s, y_s, Z_q, y_q, train_loss, predictive_val_loss, is_functional_call)

    def forward_query_loader(self, s_data, q_loader, train_loss=None, s_label=None, q_pred_adj=False,
                             predictive_val_loss: bool = False, is_functional_call: bool = False):
        """Evaluation: sigmoid of the posterior mean for every query batch (adkf_model.py:136-160)."""
        assert not self.training and train_loss is None
        with torch.no_grad():
            Z_s = self._encode(s_data)
            y_s = self._convert_bool_labels(s_label)
            self.gp_model.set_train_data(inputs=Z_s, targets=y_s, strict=False)
            means, labels = [], []
            for q_data in q_loader:
                q_data = q_data.to(Z_s.device)
                labels.append(q_data.y)
                means.append(self._posterior(Z_s, y_s, self._encode(q_data)).mean)
        return torch.sigmoid(torch.cat(means, 0)), torch.cat(labels, 0)
