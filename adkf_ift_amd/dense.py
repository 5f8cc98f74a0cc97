"""Dense layers (torch.nn.Linear of the feature extractor: fs_mol/modules/gnn.py:477-515, fs_mol/modules/graph_readout.py) with their
FP32 products on the BF16 matrix pipe: ``adkf_dense_forward`` (csrc/dense_x3.h) for the forward product and for the product with
respect to the input; the weight gradient (a reduction over all rows: another operand layout) stays a library GEMM.

``linear(x, weight, bias)`` is ``F.linear`` for every shape; it takes the HIP kernel where that pays - float32 CUDA tensors, at least
``MIN_ROWS`` rows, contraction and output widths that are multiples of 32 and at least ``MIN_K`` / ``MIN_N`` - and ``F.linear`` otherwise.
``ADKF_X3_DENSE=0`` (read at import) sends everything to ``F.linear`` for A/B runs."""
import ctypes as C
import os

import torch
import torch.nn.functional as F

ENABLED = os.environ.get("ADKF_X3_DENSE", "1") != "0"
# the weight gradient on the BF16 pipe as well (k_dense3_tn): measured EQUAL to the library GEMM on the C3 step (51.0 - 51.1 ms either way,
# tools/r05_dense.sh), so the library product stays the default; 1 selects the kernel (fixed-order partial sums: bit-reproducible)
WEIGHT_GRAD = os.environ.get("ADKF_X3_DENSE_WGRAD", "0") == "1"
# (the fc head - 2 304 rows at C3 - measured on the kernel too: 51.1 -> 51.5 ms per step: its few row tiles do not fill the chip)
MIN_ROWS, MIN_K, MIN_N = 4096, 512, 128   # measured at C3 (tools/r05_dense.sh): 52.1 -> 50.9 ms per step with every such layer, 51.2 with the wide (>= 512) ones only


def _split(w: torch.Tensor) -> torch.Tensor:
    """[rows, K] float32 -> planes [3, rows, K] of bfloat16 bit patterns (their sum is w exactly)."""
    from . import _lib
    lib = _lib.load()
    w = w.contiguous()
    planes = torch.empty((3,) + tuple(w.shape), dtype=torch.int16, device=w.device)
    st = C.c_void_p(torch.cuda.current_stream(w.device).cuda_stream)
    _lib.check(lib.adkf_split_planes(C.c_void_p(w.data_ptr()), C.c_void_p(planes.data_ptr()), w.shape[0], w.shape[1], st), "adkf_split_planes")
    return planes


def _split_t(w: torch.Tensor) -> torch.Tensor:
    """[K, N] float32 (a weight as ``x @ w`` takes it) -> planes [3, N, K]: the planes of w^T without forming w^T."""
    from . import _lib
    lib = _lib.load()
    w = w.contiguous()
    K, N = w.shape
    planes = torch.empty((3, N, K), dtype=torch.int16, device=w.device)
    st = C.c_void_p(torch.cuda.current_stream(w.device).cuda_stream)
    _lib.check(lib.adkf_split_planes_t(C.c_void_p(w.data_ptr()), C.c_void_p(planes.data_ptr()), K, N, st), "adkf_split_planes_t")
    return planes


def _dense(x: torch.Tensor, planes: torch.Tensor, bias, N: int) -> torch.Tensor:
    from . import _lib
    lib = _lib.load()
    M, K = x.shape
    y = torch.empty(M, N, dtype=torch.float32, device=x.device)
    st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
    _lib.check(lib.adkf_dense_forward(C.c_void_p(x.data_ptr()), x.stride(0), C.c_void_p(planes.data_ptr()),
                                      C.c_void_p(bias.data_ptr()) if bias is not None else None, C.c_void_p(y.data_ptr()), N, M, N, K, st),
               "adkf_dense_forward")
    return y


def _weight_grad(g: torch.Tensor, x: torch.Tensor) -> torch.Tensor:
    """g^T x on the BF16 pipe (csrc/dense_x3.h::k_dense3_tnd): row ranges, partial products added in a fixed order."""
    from . import _lib
    lib = _lib.load()
    M, N = g.shape
    K = x.shape[1]
    dw = torch.empty(N, K, dtype=torch.float32, device=g.device)
    need = lib.adkf_dense_weight_grad_scratch_bytes(M, N, K)
    scratch = torch.empty(need, dtype=torch.uint8, device=g.device)
    st = C.c_void_p(torch.cuda.current_stream(g.device).cuda_stream)
    _lib.check(lib.adkf_dense_weight_grad(C.c_void_p(g.data_ptr()), g.stride(0), C.c_void_p(x.data_ptr()), x.stride(0), C.c_void_p(dw.data_ptr()),
                                          M, N, K, C.c_void_p(scratch.data_ptr()), need, st), "adkf_dense_weight_grad")
    return dw


def _rows_ok(t: torch.Tensor) -> bool:
    return t.stride(1) == 1 and t.stride(0) % 4 == 0 and t.data_ptr() % 16 == 0


class _X3Linear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        x2 = x if _rows_ok(x) else x.contiguous()
        b = bias.contiguous() if bias is not None else None
        y = _dense(x2, _split(weight), b, weight.shape[0])
        ctx.save_for_backward(x2, weight)
        ctx.has_bias = bias is not None
        return y

    @staticmethod
    @torch.autograd.function.once_differentiable   # (library calls inside: a second differentiation must fail loudly, not silently drop terms)
    def backward(ctx, g):
        x, weight = ctx.saved_tensors
        g2 = g if _rows_ok(g) else g.contiguous()
        gx = gw = gb = None
        if ctx.needs_input_grad[0]:
            # dL/dx = g W: the same kernel on the planes of W^T (contraction over the output width) where that is long enough to pay
            if weight.shape[0] >= MIN_K and weight.shape[0] % 32 == 0 and weight.shape[1] >= MIN_N:
                gx = _dense(g2, _split(weight.t()), None, weight.shape[1])
            else:
                gx = g2 @ weight
        if ctx.needs_input_grad[1]:
            gw = _weight_grad(g2, x) if WEIGHT_GRAD else g2.t() @ x
        if ctx.has_bias and ctx.needs_input_grad[2]:
            gb = g2.sum(0)
        return gx, gw, gb


def takes_hip_kernel(x: torch.Tensor, weight: torch.Tensor) -> bool:
    N, K = weight.shape
    return (ENABLED and x.is_cuda and x.dtype == torch.float32 and weight.dtype == torch.float32 and x.dim() == 2 and x.shape[0] >= MIN_ROWS
            and K >= MIN_K and K % 32 == 0 and N % 32 == 0 and N >= MIN_N)


def linear(x: torch.Tensor, weight: torch.Tensor, bias=None) -> torch.Tensor:
    if takes_hip_kernel(x, weight):
        return _X3Linear.apply(x, weight, bias)
    return F.linear(x, weight, bias)
