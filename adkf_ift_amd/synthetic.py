"""Synthetic few-shot tasks of SURVEY.md section 8(d).

Per task ``t`` (``torch.Generator(seed = 1234 + t)``): ``X_s [N,d], X_q [N_q,d] ~ N(0,1)`` float32,
features ``Z = X W / sqrt(d)`` with a shared ``W [d,d] ~ N(0,1)`` (seed 0) playing the role of the
outer parameters theta (the stand-in for the GNN+fc of fs_mol/models/adaptive_dkt.py:141-160), labels
drawn from ``f ~ GP(0, RBF(l = sqrt(d)))`` on ``[Z_s; Z_q]`` in float64: ``y = sign(f)`` (classification,
+-1 as fs_mol/models/adaptive_dkt.py:207-209) or ``f + 0.1 eps`` standardised with the support
statistics (regression, fs_mol/data/dkt.py:91-97).  Everything is generated on the CPU so that the
same tasks exist on every rank / box regardless of the device RNG.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch


@dataclass
class SyntheticTasks:
    X_s: torch.Tensor  # [T, N, d]  float32
    X_q: torch.Tensor  # [T, Nq, d] float32
    y_s: torch.Tensor  # [T, N]     float32
    y_q: torch.Tensor  # [T, Nq]    float32
    W: torch.Tensor    # [d, d]     float32 (theta)

    def features(self, W: Optional[torch.Tensor] = None):
        W = self.W if W is None else W
        d = W.shape[0]
        return (self.X_s @ W) / math.sqrt(d), (self.X_q @ W) / math.sqrt(d)

    def to(self, device):
        return SyntheticTasks(*(t.to(device) for t in (self.X_s, self.X_q, self.y_s, self.y_q, self.W)))


def make_outer_weight(d: int) -> torch.Tensor:
    g = torch.Generator().manual_seed(0)
    return torch.randn(d, d, generator=g, dtype=torch.float32)


def make_tasks(T: int, N: int, d: int, N_q: Optional[int] = None, regression: bool = False,
               first_task: int = 0) -> SyntheticTasks:
    N_q = N if N_q is None else N_q
    W = make_outer_weight(d)
    Xs, Xq, ys, yq = [], [], [], []
    for t in range(first_task, first_task + T):
        g = torch.Generator().manual_seed(1234 + t)
        x_s = torch.randn(N, d, generator=g, dtype=torch.float32)
        x_q = torch.randn(N_q, d, generator=g, dtype=torch.float32)
        z = (torch.cat([x_s, x_q]) @ W / math.sqrt(d)).double()
        d2 = torch.cdist(z, z) ** 2
        K = torch.exp(-0.5 * d2 / d) + 1e-8 * torch.eye(N + N_q, dtype=torch.float64)
        L = torch.linalg.cholesky(K)
        f = L @ torch.randn(N + N_q, generator=g, dtype=torch.float64)
        if regression:
            f = f + 0.1 * torch.randn(N + N_q, generator=g, dtype=torch.float64)
            mu, sd = f[:N].mean(), f[:N].std()
            y = (f - mu) / sd
        else:
            y = torch.where(f >= 0, torch.ones_like(f), -torch.ones_like(f))
        Xs.append(x_s)
        Xq.append(x_q)
        ys.append(y[:N].float())
        yq.append(y[N:].float())
    return SyntheticTasks(torch.stack(Xs), torch.stack(Xq), torch.stack(ys), torch.stack(yq), W)


_X3_DW = __import__("os").environ.get("ADKF_X3_DW", "1") != "0"   # A/B: 0 keeps the chunked bmm + sum
_X3_FWD = __import__("os").environ.get("ADKF_X3_FWD", "1") != "0"   # A/B: 0 keeps the forward product on the library GEMM


class _ChunkedLinear(torch.autograd.Function):
    """Z = X W for rows X that already carry the 1 / sqrt(d) of the feature map (the inputs are constants: scaling them once
    replaces an element-wise pass over W before the forward product and one over dW behind the backward one, two launches
    per meta-step).  Backward dW = X^T dZ has a 256 x 256 output and a reduction over all T*(N+Nq) rows: as one GEMM it
    leaves most CUs idle (measured 265 us on MI355X at C2), as a 32-way chunked bmm + sum it takes 140 us."""

    @staticmethod
    def forward(ctx, X2, W, chunks, planes=None):
        ctx.save_for_backward(X2)
        ctx.chunks = chunks
        if _X3_FWD and X2.is_cuda and X2.dtype == torch.float32 and W.shape[0] in (64, 128, 256) and X2.shape[0] >= 32768:
            # the forward product on the BF16 pipe too (csrc/dense_x3.h::k_dense3_sk: FP32-accurate, 61 us against the library GEMM's 69
            # inside the C2 step); the planes of W^T are written by one small launch per step (adkf_split_planes_t)
            # - or none at all when the optimiser's step wrote them with the weights (ClipAdam.attach_planes: ``planes``)
            from .dense import _dense, _split_t
            return _dense(X2, planes if planes is not None else _split_t(W.detach()), None, W.shape[1])
        return X2 @ W

    @staticmethod
    def backward(ctx, g):
        (X2,) = ctx.saved_tensors
        ch = ctx.chunks
        R, d = X2.shape
        g = g.reshape(R, -1)
        if _X3_DW and X2.is_cuda and X2.dtype == torch.float32 and g.dtype == torch.float32 and R >= 4096:
            # the same row-range scheme in one kernel, on the BF16 matrix pipe at FP32 accuracy (csrc/dense_x3.h::k_dense3_tnd)
            from .dense import _weight_grad
            return None, _weight_grad(X2.contiguous(), g.contiguous()), None, None
        if R % ch or ch == 1:
            return None, X2.t() @ g, None, None
        part = torch.bmm(X2.view(ch, R // ch, d).transpose(1, 2), g.view(ch, R // ch, -1))
        return None, part.sum(0), None, None


class LinearFeatureMap:
    """The synthetic stand-in for the GNN + fc head: Z = (X / sqrt(d)) W for support and query rows in ONE matmul.
    ``__call__()`` returns the stacked features ``[2, T, N, d]`` (support, query) when N == Nq, else a pair."""

    def __init__(self, X_s: torch.Tensor, X_q: torch.Tensor, W: torch.Tensor, chunks: int = 32):
        self.W, self.chunks = W, chunks
        self.optimizer = None
        self.c = 1.0 / math.sqrt(W.shape[0])
        self.same = X_s.shape == X_q.shape
        if self.same:
            self.X = (torch.stack([X_s, X_q]) * self.c).contiguous()
        else:
            self.X_s, self.X_q = X_s, X_q

    def planes_from(self, optimizer) -> None:
        """``optimizer`` (a ``trainer.ClipAdam`` over ``W``) writes the bfloat16 planes of the updated weights in the launch that updates
        them (``adkf_clip_adam_step_one``); the forward product then needs no split launch of its own.  Any step that did not write them
        (the first one, another optimiser form, an in-place change of ``W``) falls back to splitting: ``ClipAdam.fresh_planes``."""
        W = self.W
        if W.is_cuda and W.dtype == torch.float32 and W.dim() == 2 and W.shape[0] % 64 == 0 and W.shape[1] % 64 == 0 and W.numel() <= optimizer.ONE_MAX:
            optimizer.attach_planes(W, torch.empty(3, W.shape[1], W.shape[0], dtype=torch.int16, device=W.device))
            self.optimizer = optimizer

    def __call__(self):
        if self.same:
            sh = self.X.shape
            planes = self.optimizer.fresh_planes(self.W) if self.optimizer is not None else None
            return _ChunkedLinear.apply(self.X.view(-1, sh[-1]), self.W, self.chunks, planes).view(sh[0], sh[1], sh[2], -1)
        return (self.X_s @ self.W) * self.c, (self.X_q @ self.W) * self.c


# ---------------------------------------------------------------------------------------------------------------------
# Synthetic MOLECULAR tasks (config C3 and the meta-test protocol): random graphs in the FSMolBatch layout
# (fs_mol/data/fsmol_batcher.py:22-54: 32 node features, 3 edge types) + count fingerprints + descriptors.  FS-Mol itself is
# not available offline; these have its shapes (15-35 heavy atoms, a backbone of single bonds plus a few other bonds).
# ---------------------------------------------------------------------------------------------------------------------
def random_molecules(n: int, gen: torch.Generator, nodes=(15, 35)):
    # (the order in which the generator is consumed is part of the benchmark definition: the recorded library-GEMM choices of
    # gemm_tuning.py are keyed by exact shapes, node count included)
    from .meta_batch import MoleculeFeatures

    feats, n2g, adj = [], [], [[], [], []]
    v0 = 0
    for gi in range(n):
        k = int(torch.randint(nodes[0], nodes[1], (1,), generator=gen))
        feats.append(torch.randn(k, 32, generator=gen))
        n2g += [gi] * k
        adj[0].append(torch.stack([torch.arange(k - 1), torch.arange(1, k)], 1) + v0)          # a backbone of single bonds
        for t in (1, 2):
            e = int(torch.randint(0, 4, (1,), generator=gen))
            if e:
                adj[t].append(torch.randint(0, k, (e, 2), generator=gen) + v0)
        v0 += k
    adj = [torch.cat(a) if a else torch.zeros(0, 2, dtype=torch.long) for a in adj]
    return MoleculeFeatures(torch.cat(feats), adj, torch.tensor(n2g), n, torch.poisson(torch.full((n, 2048), 0.03), generator=gen),
                            torch.randn(n, 42, generator=gen))


def molecular_task(support: int, query: int, gen: torch.Generator):
    from .meta_batch import DKTBatch

    s, q = random_molecules(support, gen), random_molecules(query, gen)
    return DKTBatch(s, torch.rand(support, generator=gen) > 0.5, torch.randn(support, generator=gen),
                    q, torch.rand(query, generator=gen) > 0.5, torch.randn(query, generator=gen))


def meta_test_tasks(n_tasks: int = 157, support: int = 64, seed: int = 0):
    """Tasks of the shape of the reference's only published wall-clock protocol (fs_mol/adaptive_dkt_walltime.py:100-115,
    fs_mol/utils/test_utils.py:236-350: every FS-Mol test task - 157 - at support size 64, every remaining molecule as
    query): query sizes from a seeded log-normal (median 200, clipped to [32, 2000] - FS-Mol test tasks are of this order)."""
    import numpy as np

    rng = np.random.default_rng(seed)
    sizes = np.clip(np.exp(rng.normal(np.log(200.0), 0.9, n_tasks)), 32, 2000).astype(int)
    gen = torch.Generator().manual_seed(seed)
    return [molecular_task(support, int(q), gen) for q in sizes], sizes
