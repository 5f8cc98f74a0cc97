"""Harness row H: one outer (meta) step over a meta-batch of tasks, data-parallel over ranks.

Reproduces ``ADKTModelTrainer.train_loop`` (fs_mol/utils/adaptive_dkt_utils.py:352-413) for the hot path:
per task {reinit GP params, inner fit, IFT hypergradient}, mean of the task hypergradients, clip-by-global-norm,
optimiser step - except that all tasks of the meta-batch go through the HIP library at once and the feature
extractor runs ONE forward and ONE backward per meta-batch instead of >=3 forwards and h+1 backwards per task.

Multi-GPU: tasks shard over ranks with no data-path exchange; the only collective is one
all-reduce(sum) of the flat outer gradient, after which every rank divides by the global task count, clips
(the clip must follow the all-reduce to match the reference) and applies the identical optimiser step.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


@dataclass
class MetaStepConfig:
    gp_kernel: str = "matern"              # fs_mol/utils/adaptive_dkt_utils.py:64
    use_numeric_labels: bool = False
    use_lengthscale_prior: bool = True
    ignore_grad_correction: bool = False
    clip_value: Optional[float] = 1.0      # fs_mol/adaptive_dkt_train.py --clip_value default
    inner_max_evals: int = 200
    inner_exact_evals: bool = False        # benchmark mode: exactly inner_max_evals evaluations per task
    inner_gtol: float = 1e-5
    inner_ftol: float = 1e-7


class HipGPBackend:
    """The product backend: every call lands in libadkf_gp.so.  (Tests may substitute an oracle-backed object
    with the same three methods to exercise the harness/collective logic on CPU ranks.)"""

    def init(self, Z_s, cfg: MetaStepConfig, n_s=None):
        from . import gp_ops
        phi0, priors, _ = gp_ops.init_params(Z_s, cfg.use_numeric_labels, cfg.use_lengthscale_prior, n_s=n_s)
        return phi0, priors

    def fit(self, Z_s, y_s, priors, phi0, cfg: MetaStepConfig, n_s=None, events=None):
        from . import gp_ops
        b = gp_ops.GPBatch(Z_s, y_s, priors, cfg.gp_kernel, n_s=n_s)
        phi, f, gnorm, nev, info = gp_ops.fit(b, phi0, cfg.inner_max_evals, cfg.inner_gtol, cfg.inner_ftol,
                                              cfg.inner_exact_evals, events=events)
        return phi, info

    def hypergrad(self, Z_s, y_s, Z_q, y_q, priors, phi, cfg: MetaStepConfig, n_s=None, n_q=None):
        from . import gp_ops
        b = gp_ops.GPBatch(Z_s, y_s, priors, cfg.gp_kernel, Z_q=Z_q, y_q=y_q, n_s=n_s, n_q=n_q)
        out = gp_ops.ift_hypergrad(b, phi, ignore_grad_correction=cfg.ignore_grad_correction)
        return out["f_out"], out["dZ_s"], out["dZ_q"], out["info"]


def allreduce_flat_grads(params: Sequence[torch.Tensor], group=None) -> None:
    """One all-reduce(sum) over the concatenation of all gradients (one bucket: the payload is small next to
    a meta-step and xGMI rings are per-link bound, so fewer, larger messages win)."""
    grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
    flat = torch.cat([g.reshape(-1) for g in grads])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for p, g in zip(params, grads):
        n = g.numel()
        p.grad = flat[off:off + n].view_as(g).clone()
        off += n


def meta_step(features_fn: Callable[[], Tuple[torch.Tensor, torch.Tensor]], params: List[torch.Tensor],
              optimizer: Optional[torch.optim.Optimizer], y_s: torch.Tensor, y_q: torch.Tensor, cfg: MetaStepConfig,
              backend=None, n_s=None, n_q=None, distributed: bool = False, fit_events=None, check: bool = False):
    """One outer step.  ``features_fn()`` runs the feature extractor on this rank's tasks and returns
    ``(Z_s [T,N,d], Z_q [T,Nq,d])`` attached to ``params``.  Returns per-task per-sample losses
    ``f_out / N_q`` (fs_mol/utils/adaptive_dkt_utils.py:398)."""
    backend = backend or HipGPBackend()
    if optimizer is not None:
        optimizer.zero_grad(set_to_none=True)
    Z_s, Z_q = features_fn()
    T_local = Z_s.shape[0]
    world = dist.get_world_size() if distributed else 1
    Zs_d, Zq_d = Z_s.detach(), Z_q.detach()
    # a3/a4: fresh GP parameters per task from the detached support features (adaptive_dkt.py:178-179)
    phi0, priors = backend.init(Zs_d, cfg, n_s=n_s)
    # a7: inner fit
    phi, info_fit = backend.fit(Zs_d, y_s, priors, phi0, cfg, n_s=n_s, events=fit_events)
    # a9: hypergradient at the feature level
    f_out, dZ_s, dZ_q, info = backend.hypergrad(Zs_d, y_s, Zq_d, y_q, priors, phi, cfg, n_s=n_s, n_q=n_q)
    if check:
        from . import gp_ops
        gp_ops.check_info(info_fit, "inner fit")
        gp_ops.check_info(info, "IFT hypergradient")
    # one backward through the feature extractor; task-mean over the GLOBAL meta-batch (adaptive_dkt_utils.py:402-407)
    scale = 1.0 / float(T_local * world)
    torch.autograd.backward([Z_s, Z_q], [dZ_s.to(Z_s.dtype) * scale, dZ_q.to(Z_q.dtype) * scale])
    if distributed and world > 1:
        allreduce_flat_grads(params)
    if cfg.clip_value is not None:
        torch.nn.utils.clip_grad_norm_(params, cfg.clip_value)
    if optimizer is not None:
        optimizer.step()
    nq = n_q.to(f_out.dtype) if n_q is not None else float(Z_q.shape[1])
    return f_out / nq, phi
