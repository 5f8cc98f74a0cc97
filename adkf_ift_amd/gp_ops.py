"""Batched GP operators on the HIP library: thin torch-tensor front end of include/adkf_gp.h.

All tensors live on one ROCm device, float32, contiguous.  Shapes: ``Z_s [T,N,d]``, ``y_s [T,N]``,
``Z_q [T,Nq,d]``, ``y_q [T,Nq]``, ``phi [T,h]`` (h = 3, or 2 + d for ``ard=True`` batches), ``priors [T,4]``,
optional ragged sizes ``n_s [T]``, ``n_q [T]`` (int32).  PyTorch is used for device memory and streams only; every number comes from the
hand-written kernels in ``csrc/``.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional, Tuple

import torch

from . import _lib
from ._lib import KERNEL_MATERN52, KERNEL_RBF, Batch, FitOptions

KERNELS = {"rbf": KERNEL_RBF, "RBF": KERNEL_RBF, "matern": KERNEL_MATERN52}
REUSE_DIST = 1
REUSE_INNER = 2
LG_UNFUSED = 16     # blocked path: three launches per block step (A/B; include/adkf_gp.h)
LG_FUSED = 32       # ... the fused block step whatever the size
DEFER_REFINE = 8   # adkf_fit: leave the float64 re-evaluation of ill-conditioned tasks to the next call (include/adkf_gp.h)
ARD = 4


def kernel_id(kernel) -> int:
    if isinstance(kernel, int):
        return kernel
    if kernel not in KERNELS:
        # same message shape as fs_mol/utils/gp_utils.py:43; the other kernels of that file are out of scope
        raise ValueError("[ERROR] the kernel '" + str(kernel) + "' is not supported!")
    return KERNELS[kernel]


# SciPy's L-BFGS-B default ftol (factr * eps = 2.22e-9), which is what botorch's fit_gpytorch_scipy runs with
# (SURVEY App. A7).  On an fp32 objective of magnitude ~1 it means "stop when an accepted step does not lower f at all";
# 1e-7 (one ulp) looked equivalent and was not: it stopped fits that were still creeping along the lengthscale valley.
FTOL_DEFAULT = 2.22e-9


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _f32(t: Optional[torch.Tensor], name: str) -> Optional[torch.Tensor]:
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f"{name} must live on the GPU: the GP path has no CPU fallback")
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


@dataclass
class GPBatch:
    """A meta-batch of tasks in the library's layout."""

    Z_s: torch.Tensor
    y_s: torch.Tensor
    priors: torch.Tensor
    kernel: int = KERNEL_RBF
    Z_q: Optional[torch.Tensor] = None
    y_q: Optional[torch.Tensor] = None
    n_s: Optional[torch.Tensor] = None
    n_q: Optional[torch.Tensor] = None
    flags: int = 0  # REUSE_DIST / REUSE_INNER promises for consecutive calls on this batch (include/adkf_gp.h)
    ard: bool = False  # one lengthscale per feature dimension: phi has 2 + d entries per task (ADKF_BATCH_ARD)

    def __post_init__(self):
        self.kernel = kernel_id(self.kernel)
        self.Z_s = _f32(self.Z_s, "Z_s")
        self.y_s = _f32(self.y_s, "y_s")
        self.priors = _f32(self.priors, "priors")
        self.Z_q = _f32(self.Z_q, "Z_q")
        self.y_q = _f32(self.y_q, "y_q")
        if self.Z_s.dim() != 3:
            raise ValueError("Z_s must be [T, N, d]")
        T, N, d = self.Z_s.shape
        # The library trusts the sizes of adkf_batch_t: every tensor is checked against them HERE (a [T, 3] phi on an ARD
        # batch, or labels of another meta-batch, would otherwise be read and written out of bounds on the device).
        if tuple(self.y_s.shape) != (T, N):
            raise ValueError(f"y_s must be [T, N] = [{T}, {N}], got {tuple(self.y_s.shape)}")
        if tuple(self.priors.shape) != (T, 4):
            raise ValueError(f"priors must be [T, 4] = [{T}, 4], got {tuple(self.priors.shape)}")
        if (self.Z_q is None) != (self.y_q is None) and self.Z_q is None:
            raise ValueError("y_q given without Z_q")
        if self.Z_q is not None:
            if self.Z_q.dim() != 3 or self.Z_q.shape[0] != T or self.Z_q.shape[2] != d:
                raise ValueError(f"Z_q must be [T, N_q, d] = [{T}, N_q, {d}], got {tuple(self.Z_q.shape)}")
            if self.y_q is not None and tuple(self.y_q.shape) != tuple(self.Z_q.shape[:2]):
                raise ValueError(f"y_q must be [T, N_q] = {tuple(self.Z_q.shape[:2])}, got {tuple(self.y_q.shape)}")
        for name, limit in (("n_s", N), ("n_q", self.nq)):
            v = getattr(self, name)
            if v is not None:
                if v.numel() != T:
                    raise ValueError(f"{name} must have T = {T} entries, got {v.numel()}")
                if not v.is_cuda:     # host-side sizes can be range-checked without a device synchronisation
                    lo, hi = int(v.min()), int(v.max())
                    if lo < (1 if name == "n_s" else 0) or hi > limit:
                        raise ValueError(f"{name} out of range: [{lo}, {hi}] for a padded size of {limit}")
                setattr(self, name, v.to(device=self.Z_s.device, dtype=torch.int32).contiguous().view(T))
        for name in ("y_s", "priors", "Z_q", "y_q"):
            v = getattr(self, name)
            if v is not None and v.device != self.Z_s.device:
                raise ValueError(f"{name} lives on {v.device}, Z_s on {self.Z_s.device}: one batch, one device")
        self._ws = None

    @property
    def T(self):
        return self.Z_s.shape[0]

    @property
    def ns(self):
        return self.Z_s.shape[1]

    @property
    def nq(self):
        return 0 if self.Z_q is None else self.Z_q.shape[1]

    @property
    def d(self):
        return self.Z_s.shape[2]

    @property
    def device(self):
        return self.Z_s.device

    @property
    def h(self):
        return 2 + self.d if self.ard else 3

    def c_struct(self) -> Batch:
        b = Batch()
        flags = int(self.flags) | (ARD if self.ard else 0)
        b.T, b.ns_max, b.nq_max, b.d, b.kernel, b.flags = self.T, self.ns, self.nq, self.d, self.kernel, flags
        b.n_s, b.n_q = _ptr(self.n_s), _ptr(self.n_q)
        b.Z_s, b.y_s, b.Z_q, b.y_q, b.priors = _ptr(self.Z_s), _ptr(self.y_s), _ptr(self.Z_q), _ptr(self.y_q), _ptr(self.priors)
        return b

    def workspace(self) -> Tuple[torch.Tensor, int]:
        """The caller-owned workspace of THIS batch.  It belongs to the batch object (not to a per-stream cache) because
        the REUSE_DIST / REUSE_INNER promises refer to what earlier calls on this batch left in it; torch's caching allocator
        makes the per-batch allocation cheap."""
        lib = _lib.load()
        need = (lib.adkf_workspace_bytes_ard if self.ard else lib.adkf_workspace_bytes)(self.T, self.ns, self.nq, self.d)
        if self._ws is None or self._ws.numel() < need:
            if self._ws is not None and (self.flags & (REUSE_DIST | REUSE_INNER)):
                raise RuntimeError("the workspace of this batch would have to grow while REUSE flags are set: the data they refer to would be lost")
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws, need

    def check_phi(self, phi: torch.Tensor, name: str = "phi") -> torch.Tensor:
        phi = _f32(phi, name)
        if tuple(phi.shape) != (self.T, self.h):
            raise ValueError(f"{name} must be [T, h] = [{self.T}, {self.h}] for this batch, got {tuple(phi.shape)}")
        if phi.device != self.device:
            raise ValueError(f"{name} lives on {phi.device}, the batch on {self.device}")
        return phi


def _stream(dev) -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _new(b: GPBatch, *shape, dtype=torch.float32):
    return torch.empty(*shape, dtype=dtype, device=b.device)


def check_info(info: torch.Tensor, what: str = "GP factorisation"):
    """Raises like gpytorch's NotPSDError would in the reference (SURVEY 8b error convention)."""
    lib = _lib.load()
    rc = lib.adkf_check_info(_ptr(info), info.numel(), _stream(info.device))
    if rc > 0:
        code = int(info[rc - 1].item())
        if code >= 200000:   # ADKF_INFO_CG_BASE
            raise RuntimeError(f"{what}: conjugate gradients met non-positive curvature for task {rc - 1} at iteration "
                               f"{code - 200000}: the inner Hessian is not positive definite (is the inner fit converged?)")
        where = "predictive covariance" if code >= _lib.INFO_OUTER_BASE else "K + noise*I"
        raise RuntimeError(f"{what}: matrix not positive definite for task {rc - 1} ({where}, pivot {code % _lib.INFO_OUTER_BASE})")
    if rc < 0:
        _lib.check(rc, "adkf_check_info")


def median_lengthscale(b: GPBatch) -> torch.Tensor:
    lib = _lib.load()
    l0 = _new(b, b.T)
    ws, nb = b.workspace()
    cb = b.c_struct()
    _lib.check(lib.adkf_median_lengthscale(C.byref(cb), _ptr(l0), _ptr(ws), nb, _stream(b.device)), "adkf_median_lengthscale")
    return l0


def init_params(Z_s: torch.Tensor, use_numeric_labels: bool = False, use_lengthscale_prior: bool = True,
                n_s: Optional[torch.Tensor] = None):
    """Returns (phi [T,3], priors [T,4], l0 [T]) exactly as ``reinit_gp_params`` would initialise every task."""
    lib = _lib.load()
    Z_s = _f32(Z_s, "Z_s")
    dummy = torch.zeros(Z_s.shape[0], 4, device=Z_s.device)
    b = GPBatch(Z_s, torch.zeros(Z_s.shape[:2], device=Z_s.device), dummy, KERNEL_RBF, n_s=n_s)
    phi, priors, l0 = _new(b, b.T, 3), _new(b, b.T, 4), _new(b, b.T)
    ws, nb = b.workspace()
    cb = b.c_struct()
    _lib.check(lib.adkf_init_params(C.byref(cb), int(use_numeric_labels), int(use_lengthscale_prior), _ptr(phi),
                                    _ptr(priors), _ptr(l0), _ptr(ws), nb, _stream(b.device)), "adkf_init_params")
    return phi, priors, l0


def init_params_batch(b: GPBatch, use_numeric_labels: bool = False, use_lengthscale_prior: bool = True):
    """Like ``init_params`` but on an existing batch: fills ``b.priors`` in place and returns (phi0, l0).  The
    squared distances it computes stay in the workspace, so the caller may set ``b.flags |= REUSE_DIST`` for the
    following ``fit`` / ``ift_hypergrad`` calls on the same batch."""
    lib = _lib.load()
    phi, l0 = _new(b, b.T, b.h), _new(b, b.T)
    ws, nb = b.workspace()
    cb = b.c_struct()
    _lib.check(lib.adkf_init_params(C.byref(cb), int(use_numeric_labels), int(use_lengthscale_prior), _ptr(phi),
                                    _ptr(b.priors), _ptr(l0), _ptr(ws), nb, _stream(b.device)), "adkf_init_params")
    return phi, l0


def mll_value_grad(b: GPBatch, phi: torch.Tensor, want_grad_phi=True, want_dZ=False):
    lib = _lib.load()
    phi = b.check_phi(phi)
    f = _new(b, b.T)
    g = _new(b, b.T, b.h) if want_grad_phi else None
    dZ = _new(b, b.T, b.ns, b.d) if want_dZ else None
    info = _new(b, b.T, dtype=torch.int32)
    ws, nb = b.workspace()
    cb = b.c_struct()
    _lib.check(lib.adkf_mll_value_grad(C.byref(cb), _ptr(phi), _ptr(f), _ptr(g), _ptr(dZ), _ptr(info), _ptr(ws), nb,
                                       _stream(b.device)), "adkf_mll_value_grad")
    return f, g, dZ, info


def fit(b: GPBatch, phi0: torch.Tensor, max_evals: int = 200, gtol: float = 1e-5, ftol: float = FTOL_DEFAULT,
        exact_evals: bool = False, events: Optional[Tuple[torch.cuda.Event, torch.cuda.Event]] = None,
        inplace: bool = False):
    """Batched inner optimisation; returns (phi*, f_final, gnorm, n_evals, info).  ``events`` = a pair of
    already-created timing events recorded right around the optimiser kernel (bench.py's roofline clock).
    ``inplace``: the library optimises ``phi0`` where it lies (the meta-step's freshly initialised parameters have no
    other reader: one copy kernel less per step); otherwise ``phi0`` is left untouched."""
    lib = _lib.load()
    phi = b.check_phi(phi0, "phi0")
    if not inplace:
        phi = phi.clone()
    f, gn = _new(b, b.T), _new(b, b.T)
    ne, info = _new(b, b.T, dtype=torch.int32), _new(b, b.T, dtype=torch.int32)
    opt = FitOptions(int(max_evals), int(exact_evals), float(gtol), float(ftol),
                     events[0].cuda_event if events else None, events[1].cuda_event if events else None)
    ws, nb = b.workspace()
    cb = b.c_struct()
    _lib.check(lib.adkf_fit(C.byref(cb), _ptr(phi), C.byref(opt), _ptr(f), _ptr(gn), _ptr(ne), _ptr(info), _ptr(ws), nb,
                            _stream(b.device)), "adkf_fit")
    return phi, f, gn, ne, info


def predict(b: GPBatch, phi: torch.Tensor, want_var=True, want_cov=False):
    lib = _lib.load()
    phi = b.check_phi(phi)
    mean = _new(b, b.T, b.nq)
    var = _new(b, b.T, b.nq) if want_var else None
    cov = _new(b, b.T, b.nq, b.nq) if want_cov else None
    info = _new(b, b.T, dtype=torch.int32)
    ws, nb = b.workspace()
    cb = b.c_struct()
    _lib.check(lib.adkf_predict(C.byref(cb), _ptr(phi), _ptr(mean), _ptr(var), _ptr(cov), _ptr(info), _ptr(ws), nb,
                                _stream(b.device)), "adkf_predict")
    return mean, var, cov, info


def double_path_tasks(b: GPBatch) -> torch.Tensor:
    """[T] int32: 1 where the last ``ift_hypergrad`` / ``outer_nll_value_grad`` on this batch sent the task through the float64
    path (ill-conditioned tasks, csrc/refine64.h).  Diagnostic."""
    lib = _lib.load()
    flagged = _new(b, b.T, dtype=torch.int32)
    ws, nb = b.workspace()
    cb = b.c_struct()
    _lib.check(lib.adkf_double_path_tasks(C.byref(cb), _ptr(flagged), _ptr(ws), nb, _stream(b.device)), "adkf_double_path_tasks")
    return flagged


def outer_nll_value_grad(b: GPBatch, phi: torch.Tensor, want_grads=True):
    lib = _lib.load()
    phi = b.check_phi(phi)
    f = _new(b, b.T)
    g = _new(b, b.T, b.h) if want_grads else None
    dZs = _new(b, b.T, b.ns, b.d) if want_grads else None
    dZq = _new(b, b.T, b.nq, b.d) if want_grads else None
    info = _new(b, b.T, dtype=torch.int32)
    ws, nb = b.workspace()
    cb = b.c_struct()
    _lib.check(lib.adkf_outer_nll_value_grad(C.byref(cb), _ptr(phi), _ptr(f), _ptr(g), _ptr(dZs), _ptr(dZq), _ptr(info),
                                             _ptr(ws), nb, _stream(b.device)), "adkf_outer_nll_value_grad")
    return f, g, dZs, dZq, info


def ift_hypergrad(b: GPBatch, phi: torch.Tensor, ignore_grad_correction=False, ignore_direct_grad=False, out_dZ=None,
                  cg_maxiter: Optional[int] = None, cg_tol: float = 1e-6):
    """Returns dict(f_out, dZ_s, dZ_q, g_phi, v, H, info): the IFT hypergradient at the feature level.
    ``out_dZ = (dZ_s, dZ_q)``: optional preallocated contiguous float32 outputs (e.g. two halves of one buffer).
    ARD batches: H is None (never formed), ``cg_iters`` reports the conjugate-gradient iterations per task."""
    lib = _lib.load()
    phi = b.check_phi(phi)
    flags = (_lib.IGNORE_GRAD_CORRECTION if ignore_grad_correction else 0) | (_lib.IGNORE_DIRECT_GRAD if ignore_direct_grad else 0)
    if out_dZ is not None:
        dZ_s, dZ_q = out_dZ
        assert dZ_s.is_contiguous() and dZ_q.is_contiguous() and dZ_s.dtype == torch.float32 and dZ_q.dtype == torch.float32
        assert dZ_s.shape == b.Z_s.shape and dZ_q.shape == b.Z_q.shape
    else:
        dZ_s, dZ_q = _new(b, b.T, b.ns, b.d), _new(b, b.T, b.nq, b.d)
    ws, nb = b.workspace()
    cb = b.c_struct()
    if b.ard:
        out = dict(f_out=_new(b, b.T), dZ_s=dZ_s, dZ_q=dZ_q, g_phi=_new(b, b.T, b.h), v=_new(b, b.T, b.h), H=None,
                   info=_new(b, b.T, dtype=torch.int32), cg_iters=_new(b, b.T, dtype=torch.int32))
        _lib.check(lib.adkf_ift_hypergrad_cg(C.byref(cb), _ptr(phi), flags, int(cg_maxiter or 48), float(cg_tol), _ptr(out["f_out"]),
                                             _ptr(dZ_s), _ptr(dZ_q), _ptr(out["g_phi"]), _ptr(out["v"]), _ptr(out["cg_iters"]),
                                             _ptr(out["info"]), _ptr(ws), nb, _stream(b.device)), "adkf_ift_hypergrad_cg")
        return out
    out = dict(f_out=_new(b, b.T), dZ_s=dZ_s, dZ_q=dZ_q, g_phi=_new(b, b.T, 3),
               v=_new(b, b.T, 3), H=_new(b, b.T, 9), info=_new(b, b.T, dtype=torch.int32))
    _lib.check(lib.adkf_ift_hypergrad(C.byref(cb), _ptr(phi), flags, _ptr(out["f_out"]), _ptr(out["dZ_s"]), _ptr(out["dZ_q"]),
                                      _ptr(out["g_phi"]), _ptr(out["v"]), _ptr(out["H"]), _ptr(out["info"]), _ptr(ws), nb,
                                      _stream(b.device)), "adkf_ift_hypergrad")
    out["H"] = out["H"].view(b.T, 3, 3)
    return out
