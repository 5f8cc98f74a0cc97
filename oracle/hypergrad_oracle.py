"""CPU restatement of the reference's dense IFT hypergradient.  TEST INFRASTRUCTURE ONLY
(imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product).

Follows /root/reference/fs_mol/utils/cauchy_hypergradient.py:5-163 step for step:
  (1) dense inner Hessian with ``autograd.functional.hessian``            (:43-66)
  (2) dense mixed partials d2 f_in / d phi d theta with nested ``jacobian`` (:78-107)
  (3) ``f_outer(...).backward()``                                          (:120-121)
  (4) ``v = linalg.solve(H, d f_out / d phi)``                             (:129-136)
  (5) ``theta.grad -= tensordot(v, mixed)``                                (:139-161)
so that its cost profile (h double-backward passes through the feature extractor) is the
reference's, which is what bench.py's ``cpu_baseline`` times.  Pinned against the reference's own
file (imported by path, this container only) in tests/golden/make_golden.py and
tests/test_hypergrad_known_answers.py, and against the known answers of
/root/reference/test_hypergrad.ipynb.
"""
from __future__ import annotations

from typing import Callable, Sequence, Tuple

import torch
from torch.autograd.functional import hessian, jacobian


def _numel(ts: Sequence[torch.Tensor]) -> int:
    return sum(t.numel() for t in ts)


def dense_ift_hypergradient(f_outer: Callable, f_inner: Callable, params_outer: Tuple[torch.Tensor, ...],
                            params_inner: Tuple[torch.Tensor, ...], ignore_grad_correction: bool = False,
                            ignore_direct_grad: bool = False):
    for t in (*params_outer, *params_inner):
        t.grad = None
    h = _numel(params_inner)
    ref = params_inner[0]

    if not ignore_grad_correction:
        blocks = hessian(lambda *p: f_inner(params_outer, p), params_inner)
        H = ref.new_zeros(h, h)
        r0 = 0
        for i, pi in enumerate(params_inner):
            c0 = 0
            for j, pj in enumerate(params_inner):
                H[r0:r0 + pi.numel(), c0:c0 + pj.numel()] = blocks[i][j].reshape(pi.numel(), pj.numel())
                c0 += pj.numel()
            r0 += pi.numel()

        def grad_inner(p_out):
            return jacobian(lambda *p: f_inner(p_out, p), params_inner, create_graph=True)

        mixed_blocks = jacobian(lambda *po: grad_inner(po), params_outer)
        mixed = []
        for j, po in enumerate(params_outer):
            J = ref.new_zeros(h, *po.shape)
            r0 = 0
            for i, pi in enumerate(params_inner):
                J[r0:r0 + pi.numel()] = mixed_blocks[i][j].reshape(pi.numel(), *po.shape)
                r0 += pi.numel()
            mixed.append(J)

    value = f_outer(params_outer, params_inner)
    value.backward()

    if not ignore_grad_correction:
        g = torch.cat([p.grad.reshape(-1) for p in params_inner])
        v = torch.linalg.solve(H, g)

    for j, po in enumerate(params_outer):
        if po.grad is None:
            po.grad = torch.zeros_like(po)
        if ignore_direct_grad:
            po.grad.zero_()
        if not ignore_grad_correction:
            po.grad -= torch.tensordot(v, mixed[j], dims=1)
    return value
