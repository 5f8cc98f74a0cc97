"""``torch.nn.functional.linear`` (+ ReLU / leaky-ReLU) of the feature extractor on the library's fp32 MFMA GEMM
(csrc/dense.h, ``adkf_dense_forward`` / ``adkf_dense_backward``).

Used by the BOOM layer and message-output projection of the GNN blocks (fs_mol/modules/gnn.py:497-513), the read-out MLPs
(fs_mol/modules/graph_readout.py:119-177) and the fc head (fs_mol/models/adaptive_dkt.py:61-65) when the tensors are float32
on the GPU.  Anything else (CPU tensors, float64 - the oracle-comparison tests) takes the plain PyTorch formulation of the
same arithmetic; the GP path has no such alternative."""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch
import torch.nn.functional as F

ACT_NONE, ACT_RELU, ACT_LEAKY = 0, 1, 2


def _torch_act(y: torch.Tensor, act: int) -> torch.Tensor:
    return F.relu(y) if act == ACT_RELU else (F.leaky_relu(y) if act == ACT_LEAKY else y)


class _HipLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, act):
        from . import _lib
        lib = _lib.load()
        x2 = x.reshape(-1, x.shape[-1]).contiguous()
        w = weight.contiguous()
        M, K = x2.shape
        N = w.shape[0]
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        b = bias.contiguous() if bias is not None else None
        _lib.check(lib.adkf_dense_forward(C.c_void_p(x2.data_ptr()), C.c_void_p(w.data_ptr()), C.c_void_p(b.data_ptr()) if b is not None else None,
                                          M, N, K, int(act), C.c_void_p(y.data_ptr()), st), "adkf_dense_forward")
        ctx.save_for_backward(x2, w, y if act else None)
        ctx.act, ctx.has_bias, ctx.in_shape = int(act), bias is not None, x.shape
        return y.view(*x.shape[:-1], N)

    @staticmethod
    def backward(ctx, dy):
        from . import _lib
        lib = _lib.load()
        x2, w, yact = ctx.saved_tensors
        M, K = x2.shape
        N = w.shape[0]
        dy2 = dy.reshape(M, N).contiguous()
        need_x, need_w, need_b = ctx.needs_input_grad[0], ctx.needs_input_grad[1], ctx.has_bias and ctx.needs_input_grad[2]
        dx = torch.empty_like(x2) if need_x else None
        dw = torch.zeros_like(w) if need_w else None
        db = torch.zeros(N, dtype=torch.float32, device=w.device) if need_b else None
        ptr = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        st = C.c_void_p(torch.cuda.current_stream(x2.device).cuda_stream)
        _lib.check(lib.adkf_dense_backward(ptr(x2), ptr(w), ptr(yact), ptr(dy2), M, N, K, ctx.act, ptr(dx), 0, ptr(dw), ptr(db), st),
                   "adkf_dense_backward")
        return (dx.view(ctx.in_shape) if dx is not None else None), dw, db, None


def linear(x: torch.Tensor, weight: torch.Tensor, bias: Optional[torch.Tensor] = None, act: int = ACT_NONE) -> torch.Tensor:
    """act(x @ weight.T + bias) - on the HIP library for float32 GPU tensors, plain PyTorch otherwise."""
    if x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.numel() > 0:
        return _HipLinear.apply(x, weight, bias, act)
    return _torch_act(F.linear(x, weight, bias), act)
