"""IFT hypergradient operators with the reference's surface.

``cauchy_hypergradient`` / ``cauchy_hypergradient_jvp`` keep the signature, return value and side effects of
fs_mol/utils/cauchy_hypergradient.py:5-163 and fs_mol/utils/cauchy_hypergradient_jvp.py:5-156:

    f_value = cauchy_hypergradient(f_outer, f_inner, params_outer, params_inner, device,
                                   ignore_grad_correction=False, sanity_checks=False, ignore_direct_grad=False)

sets ``p.grad`` of every outer parameter to  d f_out/d theta - (d f_out/d phi) H^-1 d2 f_in/(d phi d theta)  and
leaves ``d f_out/d phi`` in the inner parameters' ``.grad``.

Two execution paths, same numbers:

* callables built by ``adkf_ift_amd.models`` (``GPTaskLoss`` objects) take the FUSED path: the support/query
  features are computed once, the whole GP part (f_out, its gradients, the 3x3 Hessian, the solve and the
  mixed-partial VJP) runs in the HIP library (``adkf_ift_hypergrad``), and ONE ordinary backward through the
  feature extractor turns dL/dZ into theta.grad.  The reference needs >= 3 forwards and h+1 (double-)backward
  passes of the feature extractor for the same result.
* arbitrary callables (the toy problems of /root/reference/test_hypergrad.ipynb, or any torch function) take
  the GENERIC path on whatever device their tensors live on: H from h reverse passes over grad_phi f_in, and the
  correction as ONE vector-Jacobian product d(v^T grad_phi f_in)/d theta instead of the reference's dense
  h x |theta| Jacobian (``cauchy_hypergradient``) or as a forward-over-reverse directional derivative
  (``cauchy_hypergradient_jvp``).
"""
from __future__ import annotations

from typing import Callable, Sequence, Tuple

import torch


def _clear(*groups: Sequence[torch.Tensor]) -> None:
    for grp in groups:
        for t in grp:
            t.grad = None


def _flat(ts: Sequence[torch.Tensor]) -> torch.Tensor:
    return torch.cat([t.reshape(-1) for t in ts])


def _inner_hessian(f_inner, params_outer, params_inner, device) -> Tuple[torch.Tensor, torch.Tensor]:
    """Dense H = d2 f_in / d phi2 (h x h) and the graph-attached flat gradient it came from."""
    val = f_inner(params_outer, params_inner)
    g = torch.autograd.grad(val, params_inner, create_graph=True, allow_unused=True)
    g = [gi if gi is not None else torch.zeros_like(p) for gi, p in zip(g, params_inner)]
    gflat = _flat(g)
    h = gflat.numel()
    H = torch.zeros(h, h, device=device, dtype=gflat.dtype)
    for i in range(h):
        if not gflat[i].requires_grad:
            continue  # this gradient entry is constant in phi: zero Hessian row
        row = torch.autograd.grad(gflat[i], params_inner, retain_graph=True, allow_unused=True)
        H[i] = _flat([r if r is not None else torch.zeros_like(p) for r, p in zip(row, params_inner)])
    return H, gflat


def _check_hessian(H: torch.Tensor) -> None:
    logabsdet = torch.linalg.slogdet(H).logabsdet
    if logabsdet < -2:
        print(f"WARNING: determinant seems low ({logabsdet:.5g}). perhaps Hessian is not invertible?")
    assert logabsdet.item() > -10.0


def _finish(params_outer, corrections, ignore_direct_grad, ignore_grad_correction, device):
    for j, p in enumerate(params_outer):
        if p.grad is None:
            p.grad = torch.zeros_like(p).to(device)
            p.grad.requires_grad_(False)
        if ignore_direct_grad:
            p.grad.zero_()
        if not ignore_grad_correction:
            c = corrections[j]
            if c is not None:
                assert p.grad.shape == c.shape
                p.grad -= c.detach()


def _fused_applicable(f_outer, f_inner) -> bool:
    from .models import GPTaskLoss

    return isinstance(f_outer, GPTaskLoss) and isinstance(f_inner, GPTaskLoss) and f_outer.task is f_inner.task


def cauchy_hypergradient(f_outer: Callable, f_inner: Callable, params_outer: Tuple[torch.Tensor, ...],
                         params_inner: Tuple[torch.Tensor, ...], device, ignore_grad_correction: bool = False,
                         sanity_checks: bool = False, ignore_direct_grad: bool = False):
    _clear(params_outer, params_inner)
    if _fused_applicable(f_outer, f_inner):
        return f_outer.task.fused_hypergradient(params_outer, params_inner, ignore_grad_correction,
                                                sanity_checks, ignore_direct_grad)
    corrections = None
    if not ignore_grad_correction:
        H, gflat = _inner_hessian(f_inner, params_outer, params_inner, device)
        if sanity_checks:
            _check_hessian(H)
            for t in (*params_outer, *params_inner):
                assert t.grad is None
    f_value = f_outer(params_outer, params_inner)
    f_value.backward()
    if not ignore_grad_correction:
        g_out = _flat([p.grad if p.grad is not None else torch.zeros_like(p) for p in params_inner]).to(H.dtype)
        v = torch.linalg.solve(H, g_out)
        # v^T d2 f_in/(d phi d theta) = d/d theta ( v . grad_phi f_in ): one reverse pass
        if gflat.requires_grad:
            corrections = torch.autograd.grad(gflat, params_outer, grad_outputs=v, allow_unused=True)
        else:
            corrections = [None] * len(params_outer)
    _finish(params_outer, corrections, ignore_direct_grad, ignore_grad_correction, device)
    return f_value


def cauchy_hypergradient_jvp(f_outer: Callable, f_inner: Callable, params_outer: Tuple[torch.Tensor, ...],
                             params_inner: Tuple[torch.Tensor, ...], device, ignore_grad_correction: bool = False,
                             sanity_checks: bool = False, ignore_direct_grad: bool = False):
    _clear(params_outer, params_inner)
    if _fused_applicable(f_outer, f_inner):
        return f_outer.task.fused_hypergradient(params_outer, params_inner, ignore_grad_correction,
                                                sanity_checks, ignore_direct_grad)
    corrections = None
    if not ignore_grad_correction:
        H, _ = _inner_hessian(f_inner, params_outer, params_inner, device)
        if sanity_checks:
            _check_hessian(H)
    f_value = f_outer(params_outer, params_inner)
    f_value.backward()
    if not ignore_grad_correction:
        g_out = _flat([p.grad if p.grad is not None else torch.zeros_like(p) for p in params_inner]).to(H.dtype)
        v = torch.linalg.solve(H, g_out)
        del H
        v_parts, off = [], 0
        for p in params_inner:
            v_parts.append(v[off:off + p.numel()].reshape(p.shape))
            off += p.numel()

        # directional derivative, along v in phi, of grad_theta f_in: forward-over-reverse
        def grad_theta(*p_in):
            val = f_inner(params_outer, p_in)
            g = torch.autograd.grad(val, params_outer, create_graph=True, allow_unused=True)
            return tuple(gi if gi is not None else torch.zeros_like(p) for gi, p in zip(g, params_outer))

        corrections = torch.autograd.functional.jvp(grad_theta, tuple(params_inner), tuple(v_parts))[1]
    _finish(params_outer, corrections, ignore_direct_grad, ignore_grad_correction, device)
    return f_value
