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

import os

import torch
import torch.distributed as dist


@dataclass
class MetaStepConfig:
    gp_kernel: str = "matern"              # fs_mol/utils/adaptive_dkt_utils.py:64
    use_numeric_labels: bool = False
    use_lengthscale_prior: bool = True
    use_ard: bool = False                  # one lengthscale per feature dimension: IFT system solved by HVP + CG
    ignore_grad_correction: bool = False
    clip_value: Optional[float] = 1.0      # fs_mol/adaptive_dkt_train.py --clip_value default
    inner_max_evals: int = 200
    inner_exact_evals: bool = False        # benchmark mode: exactly inner_max_evals evaluations per task
    inner_gtol: float = 1e-5
    inner_ftol: float = 2.22e-9            # SciPy L-BFGS-B default (factr * eps): on an fp32 objective "no decrease at all"
    uneven_shards: bool = False            # ranks hold different numbers of tasks (node-balanced shards): the global
                                           # task count rides in the gradient all-reduce (one host sync per step to read it)
    global_tasks: Optional[int] = None     # ... unless the caller states it: every rank computes the same shard plan
                                           # (meta_batch.shard_tasks_by_nodes is deterministic), so the total is known up
                                           # front and the step stays free of host synchronisation


class HipGPBackend:
    """The product backend: every call lands in libadkf_gp.so.  (Tests may substitute an oracle-backed object with
    the same ``run`` method to exercise the harness/collective logic on CPU ranks.)"""

    def run(self, Z_s, y_s, Z_q, y_q, cfg: MetaStepConfig, n_s=None, n_q=None, fit_events=None, out_dZ=None):
        """a3/a4 re-initialisation -> a7 inner fit -> a9 IFT hypergradient for every task of the meta-batch.
        One GPBatch / one workspace for the three calls, so the squared distances are built once and the
        hypergradient reuses the A^-1, alpha the fit left behind."""
        from . import gp_ops
        priors = torch.empty(Z_s.shape[0], 4, dtype=torch.float32, device=Z_s.device)
        b = gp_ops.GPBatch(Z_s, y_s, priors, cfg.gp_kernel, Z_q=Z_q, y_q=y_q, n_s=n_s, n_q=n_q, ard=cfg.use_ard)
        phi0, _ = gp_ops.init_params_batch(b, cfg.use_numeric_labels, cfg.use_lengthscale_prior)
        # (the hypergradient call below redoes ill-conditioned tasks in float64 itself: the fit need not)
        b.flags = gp_ops.REUSE_DIST | gp_ops.DEFER_REFINE
        phi, f_in, gnorm, nev, info_fit = gp_ops.fit(b, phi0, cfg.inner_max_evals, cfg.inner_gtol, cfg.inner_ftol,
                                                     cfg.inner_exact_evals, events=fit_events, inplace=True)
        b.flags = gp_ops.REUSE_DIST | gp_ops.REUSE_INNER
        out = gp_ops.ift_hypergrad(b, phi, ignore_grad_correction=cfg.ignore_grad_correction, out_dZ=out_dZ)
        return phi, out["f_out"], out["dZ_s"], out["dZ_q"], info_fit, out["info"]


class GraphedGPBackend(HipGPBackend):
    """``HipGPBackend`` behind a HIP graph: the library only enqueues kernels and memsets on the stream it is given (no
    allocation, no synchronisation), so the whole init -> fit -> hypergradient sequence is captured once per (shapes,
    configuration) and replayed with one launch per meta-step: +12 % on launch-bound shapes (64 tasks of 32 points:
    0.515 -> 0.458 ms), -2.5 % at C2 where the GPU is the bottleneck and the input copies cost more than the launches.
    (Dealing the tasks to several streams so that one chunk's latency-bound fit overlaps the others' GEMM stages was
    measured too, eagerly and inside a graph: slower at every chunk count - 1.57 / 1.66 / 1.83 / 2.89 ms for 1 / 2 / 4 / 8.)  Inputs are copied into the graph's static buffers; the returned tensors are the graph's static outputs
    (valid until the next call).  Needs equal shapes from step to step (ragged sizes go through ``n_s`` / ``n_q``)."""

    def __init__(self):
        self._graphs = {}

    def run(self, Z_s, y_s, Z_q, y_q, cfg: MetaStepConfig, n_s=None, n_q=None, fit_events=None, out_dZ=None):
        key = (tuple(Z_s.shape), tuple(Z_q.shape), n_s is not None, n_q is not None, Z_s.device.index,
               tuple(sorted(vars(cfg).items())))
        entry = self._graphs.get(key)
        if entry is None:
            static_in = [torch.empty_like(a) if a is not None else None for a in (Z_s, y_s, Z_q, y_q, n_s, n_q)]
            for dst, src in zip(static_in, (Z_s, y_s, Z_q, y_q, n_s, n_q)):
                if dst is not None:
                    dst.copy_(src)
            side = torch.cuda.Stream(device=Z_s.device)
            side.wait_stream(torch.cuda.current_stream(Z_s.device))
            with torch.cuda.stream(side):   # warm-up outside the capture: lazy one-time initialisation (function attributes, pools)
                HipGPBackend.run(self, static_in[0], static_in[1], static_in[2], static_in[3], cfg, static_in[4], static_in[5])
            torch.cuda.current_stream(Z_s.device).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = HipGPBackend.run(self, static_in[0], static_in[1], static_in[2], static_in[3], cfg, static_in[4], static_in[5])
            entry = (graph, static_in, static_out)
            self._graphs[key] = entry
        graph, static_in, static_out = entry
        for dst, src in zip(static_in, (Z_s, y_s, Z_q, y_q, n_s, n_q)):
            if dst is not None:
                dst.copy_(src)
        graph.replay()
        phi, f_out, dZ_s, dZ_q, info_fit, info = static_out
        if out_dZ is not None:
            out_dZ[0].copy_(dZ_s)
            out_dZ[1].copy_(dZ_q)
            dZ_s, dZ_q = out_dZ
        return phi, f_out, dZ_s, dZ_q, info_fit, info


def allreduce_flat_grads(params: Sequence[torch.Tensor], group=None, local_tasks: Optional[int] = None) -> Optional[int]:
    """One all-reduce(sum) over the concatenation of all gradients (one bucket: the payload is small next to
    a meta-step and xGMI rings are per-link bound, so fewer, larger messages win).  ``local_tasks`` rides in the same
    message as one extra element and comes back as the GLOBAL task count, so shards balanced by node count (unequal
    task counts per rank, SURVEY 8e caveat) still divide by the right T without a second collective.  The count
    travels as three base-4096 digits, each exact in fp32 for any realistic number of ranks."""
    if len(params) == 1 and local_tasks is None and params[0].grad is not None:
        dist.all_reduce(params[0].grad, op=dist.ReduceOp.SUM, group=group)   # nothing to pack
        return None
    grads = [p.grad if p.grad is not None else torch.zeros_like(p) for p in params]
    parts = [g.reshape(-1) for g in grads]
    if local_tasks is not None:
        t = int(local_tasks)
        parts.append(torch.tensor([t % 4096, (t // 4096) % 4096, t // (4096 * 4096)], dtype=grads[0].dtype, device=grads[0].device))
    flat = torch.cat(parts)
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for p, g in zip(params, grads):
        n = g.numel()
        p.grad = flat[off:off + n].view_as(g).clone()
        off += n
    if local_tasks is None:
        return None
    d = flat[off:off + 3].double().round().tolist()
    return int(d[0] + 4096 * d[1] + 4096 * 4096 * d[2])


def _mean_and_clip_(params: Sequence[torch.Tensor], scale: float, clip_value: Optional[float]) -> None:
    """grad <- grad * scale (the task-mean: scaling |theta| numbers once is cheaper than scaling both dZ tensors), then
    clip-by-global-norm (fs_mol/utils/adaptive_dkt_utils.py:406-410).  For a handful of tensors the two are folded into
    ONE multiply by  scale * min(1, clip / (scale * |g| + 1e-6))  - ``clip_grad_norm_``'s foreach norm alone costs 42 us
    on a single 256 x 256 parameter; large parameter sets take torch's batched implementation."""
    grads = [p.grad for p in params if p.grad is not None]
    if not grads:
        return
    if clip_value is None:
        torch._foreach_mul_(grads, scale)
        return
    if len(grads) > 4:
        torch._foreach_mul_(grads, scale)
        torch.nn.utils.clip_grad_norm_(params, clip_value)
        return
    norms = [torch.linalg.vector_norm(g) for g in grads]
    total = norms[0] if len(norms) == 1 else torch.linalg.vector_norm(torch.stack(norms))
    coef = (clip_value / (total * scale + 1e-6)).clamp(max=1.0) * scale
    for g in grads:
        g.mul_(coef)


class ClipAdam(torch.optim.Adam):
    """``torch.optim.Adam`` whose ``clip_step(scale, clip_value)`` does {task-mean, clip-by-global-norm, Adam update}
    (fs_mol/utils/adaptive_dkt_utils.py:402-413) in the HIP library: ``adkf_grad_sumsq`` + ``adkf_clip_adam_step`` per
    tensor, i.e. 2 launches for the benchmark's single d x d parameter instead of norm + clamp + mul + a fused Adam
    whose one 64 K chunk runs on a single workgroup (70 us -> 8 us); a SINGLE tensor of at most ``ONE_MAX`` elements (the benchmark's
    case) takes ONE launch, ``adkf_clip_adam_step_one``, which can also write the bfloat16 planes of the updated weight for the next
    forward product (``attach_planes``).  State layout (``exp_avg``, ``exp_avg_sq``,
    ``step``) is torch's, so ``state_dict`` / ``load_state_dict`` interchange with ``torch.optim.Adam`` and a plain
    ``.step()`` still works.  Meant for a handful of tensors (``meta_step`` uses it when there are at most
    ``MAX_TENSORS`` parameters); a 300-tensor model is better served by torch's multi-tensor kernels."""

    MAX_TENSORS = 8
    ONE_MAX = 131072   # ADKF_CLIP_ADAM_ONE_MAX: a single tensor up to this size takes ONE launch (``adkf_clip_adam_step_one``)
    FUSE_ONE = os.environ.get("ADKF_CLIP_ADAM_ONE", "1") != "0"   # A/B: 0 keeps the two launches

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0.0):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, foreach=False, fused=False)
        self._partials = None
        self._planes = {}   # id(param) -> [planes_t, param version the planes were written for]

    def attach_planes(self, p: torch.Tensor, planes_t: torch.Tensor) -> None:
        """``p`` is a row-major weight ``w[K, N]`` used as ``x @ w`` on the BF16 matrix pipe (``dense._split_t``): let the step that
        updates it also write the three bfloat16 planes of the NEW weights into ``planes_t`` ([3, N, K] int16) - the split launch of the
        next forward pass goes away.  Only the one-launch form does this (one tensor, at most ``ONE_MAX`` elements); ``fresh_planes``
        says whether the planes belong to the weights as they are now."""
        if p.dim() != 2 or tuple(planes_t.shape) != (3, p.shape[1], p.shape[0]) or planes_t.dtype != torch.int16 or not planes_t.is_contiguous():
            raise ValueError("attach_planes: planes_t must be a contiguous int16 tensor [3, N, K] for a weight [K, N]")
        self._planes[id(p)] = [planes_t, None]

    def fresh_planes(self, p: torch.Tensor):
        """The attached planes if the last one-launch step wrote them and nothing has modified ``p`` in place since (torch's version
        counter), else None."""
        ent = self._planes.get(id(p))
        if ent is None or ent[1] is None or ent[1] != p._version:
            return None
        return ent[0]

    def _tensors(self):
        out = []
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is not None:
                    out.append((group, p))
        return out

    def _adam_state(self, p):
        state = self.state[p]
        if len(state) == 0:
            state["step"] = torch.tensor(0.0)
            state["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            state["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
        state["step"] += 1
        return state

    def clip_step(self, scale: float, clip_value: Optional[float]) -> None:
        import ctypes as C

        from . import _lib
        lib = _lib.load()
        todo = self._tensors()
        if not todo:
            return
        for group in self.param_groups:
            if group.get("amsgrad") or group.get("maximize"):
                raise RuntimeError("ClipAdam.clip_step implements plain Adam (no amsgrad, no maximize)")
        dev = todo[0][1].device
        for _, p in todo:
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.grad.is_contiguous()
                    and p.grad.dtype == torch.float32 and p.data_ptr() % 16 == 0 and p.grad.data_ptr() % 16 == 0):
                raise RuntimeError("ClipAdam.clip_step needs contiguous, 16-byte aligned float32 parameters on the GPU")
        parts = 256   # ADKF_SUMSQ_PARTS
        if self._partials is None or self._partials.numel() != parts * len(todo) or self._partials.device != dev:
            self._partials = torch.empty(parts * len(todo), dtype=torch.float32, device=dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        clip = float("inf") if clip_value is None else float(clip_value)
        if self.FUSE_ONE and len(todo) == 1 and todo[0][1].numel() <= self.ONE_MAX:
            group, p = todo[0]
            state = self._adam_state(p)
            b1, b2 = group["betas"]
            ent = self._planes.get(id(p))
            use_planes = ent is not None and p.dim() == 2 and p.shape[0] % 64 == 0 and p.shape[1] % 64 == 0 and ent[0].device == p.device
            _lib.check(lib.adkf_clip_adam_step_one(ptr(p), ptr(p.grad), ptr(state["exp_avg"]), ptr(state["exp_avg_sq"]), p.numel(), float(scale),
                                                   clip, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                                                   float(group["weight_decay"]), int(state["step"].item()),
                                                   ptr(ent[0]) if use_planes else None, p.shape[0] if use_planes else 0,
                                                   p.shape[1] if use_planes else 0, st), "adkf_clip_adam_step_one")
            if ent is not None:
                ent[1] = p._version if use_planes else None
            return
        for k, (_, p) in enumerate(todo):
            _lib.check(lib.adkf_grad_sumsq(ptr(p.grad), p.numel(), C.c_void_p(self._partials.data_ptr() + 4 * parts * k), st),
                       "adkf_grad_sumsq")
        for ent in self._planes.values():
            ent[1] = None    # (the two-launch form does not write planes)
        for group, p in todo:
            state = self._adam_state(p)
            b1, b2 = group["betas"]
            _lib.check(lib.adkf_clip_adam_step(ptr(p), ptr(p.grad), ptr(state["exp_avg"]), ptr(state["exp_avg_sq"]), p.numel(),
                                               ptr(self._partials), self._partials.numel(), float(scale), clip, float(group["lr"]),
                                               float(b1), float(b2), float(group["eps"]), float(group["weight_decay"]),
                                               int(state["step"].item()), st), "adkf_clip_adam_step")


def meta_step(features_fn: Callable[[], Tuple[torch.Tensor, torch.Tensor]], params: List[torch.Tensor],
              optimizer: Optional[torch.optim.Optimizer], y_s: torch.Tensor, y_q: torch.Tensor, cfg: MetaStepConfig,
              backend=None, n_s=None, n_q=None, distributed: bool = False, fit_events=None, check: bool = False):
    """One outer step.  ``features_fn()`` runs the feature extractor on this rank's tasks and returns
    ``(Z_s [T,N,d], Z_q [T,Nq,d])`` attached to ``params`` - or, when support and query sets have the same padded
    size, ONE stacked tensor ``[2,T,N,d]`` (support, query), in which case the library writes both cotangents into
    one buffer and a single ``backward`` call consumes it without any gather/scatter copies.  Returns per-task
    per-sample losses ``f_out / N_q`` (fs_mol/utils/adaptive_dkt_utils.py:398)."""
    backend = backend or HipGPBackend()
    if optimizer is not None:
        optimizer.zero_grad(set_to_none=True)
    feats = features_fn()
    stacked = isinstance(feats, torch.Tensor)
    if stacked:
        Z_s, Z_q = feats[0], feats[1]
    else:
        Z_s, Z_q = feats
    T_local = Z_s.shape[0]
    world = dist.get_world_size() if distributed else 1
    # a3/a4 fresh GP parameters from the detached support features (adaptive_dkt.py:178-179), a7 inner fit,
    # a9 hypergradient at the feature level
    kw = {}
    if stacked and feats.dtype == torch.float32 and feats.is_contiguous():
        dZ_all = torch.empty_like(feats)
        kw["out_dZ"] = (dZ_all[0], dZ_all[1])
    phi, f_out, dZ_s, dZ_q, info_fit, info = backend.run(Z_s.detach(), y_s, Z_q.detach(), y_q, cfg, n_s=n_s, n_q=n_q,
                                                         fit_events=fit_events, **kw)
    if check:
        from . import gp_ops
        gp_ops.check_info(info_fit, "inner fit")
        gp_ops.check_info(info, "IFT hypergradient")
    # one backward through the feature extractor; task-mean over the GLOBAL meta-batch (adaptive_dkt_utils.py:402-407)
    T_global = T_local
    if "out_dZ" in kw:
        feats.backward(dZ_all)
    else:
        torch.autograd.backward([Z_s, Z_q], [dZ_s.to(Z_s.dtype), dZ_q.to(Z_q.dtype)])
    if distributed and world > 1:
        if cfg.global_tasks is not None:
            allreduce_flat_grads(params)
            T_global = int(cfg.global_tasks)
        elif cfg.uneven_shards:
            T_global = allreduce_flat_grads(params, local_tasks=T_local)   # host reads the count: one sync per step
        else:
            allreduce_flat_grads(params)
            T_global = T_local * world
    if isinstance(optimizer, ClipAdam) and len(params) <= ClipAdam.MAX_TENSORS and params[0].is_cuda:
        optimizer.clip_step(1.0 / float(T_global), cfg.clip_value)
    else:
        _mean_and_clip_(params, 1.0 / float(T_global), cfg.clip_value)
        if optimizer is not None:
            optimizer.step()
    nq = n_q.to(f_out.dtype) if n_q is not None else float(Z_q.shape[1])
    return f_out / nq, phi
