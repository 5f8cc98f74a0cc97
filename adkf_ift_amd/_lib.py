"""ctypes binding of libadkf_gp.so (include/adkf_gp.h).  There is NO fallback: if the HIP library is not
built, importing the GP operators raises."""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libadkf_gp.so")
if os.environ.get("ADKF_LIB"):   # diagnostics: an alternative build of the same sources (e.g. with -DADKF_* experiment macros)
    LIB_PATH = os.environ["ADKF_LIB"]

KERNEL_RBF = 0
KERNEL_MATERN52 = 1
IGNORE_GRAD_CORRECTION = 1
IGNORE_DIRECT_GRAD = 2
INFO_OUTER_BASE = 100000

ERRORS = {-1: "bad argument", -2: "unsupported size (see adkf_max_points)", -3: "workspace too small",
          -4: "HIP launch failed"}


class Batch(C.Structure):
    _fields_ = [("T", C.c_int32), ("ns_max", C.c_int32), ("nq_max", C.c_int32), ("d", C.c_int32),
                ("kernel", C.c_int32), ("flags", C.c_int32),
                ("n_s", C.c_void_p), ("n_q", C.c_void_p), ("Z_s", C.c_void_p), ("y_s", C.c_void_p),
                ("Z_q", C.c_void_p), ("y_q", C.c_void_p), ("priors", C.c_void_p)]


class MsgEt(C.Structure):
    """adkf_msg_et_t"""
    _fields_ = [("src", C.c_void_p), ("tgt", C.c_void_p), ("W", C.c_void_p), ("bias", C.c_void_p), ("dW", C.c_void_p),
                ("db", C.c_void_p), ("E", C.c_int32)]


class FitOptions(C.Structure):
    _fields_ = [("max_evals", C.c_int32), ("exact_evals", C.c_int32), ("gtol", C.c_float), ("ftol", C.c_float),
                ("ev_start", C.c_void_p), ("ev_stop", C.c_void_p)]


_lib = None

# name -> (restype, argtypes); kept in one table so tests can check every declared symbol is exported
SIGNATURES = {
    "adkf_version": (C.c_char_p, []),
    "adkf_last_hip_error": (C.c_char_p, []),
    "adkf_max_points": (C.c_int, []),
    "adkf_path_info": (C.c_int, [C.c_int32, C.c_int32]),
    "adkf_workspace_bytes": (C.c_size_t, [C.c_int32] * 4),
    "adkf_median_lengthscale": (C.c_int, [C.POINTER(Batch), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "adkf_init_params": (C.c_int, [C.POINTER(Batch), C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                   C.c_void_p, C.c_size_t, C.c_void_p]),
    "adkf_mll_value_grad": (C.c_int, [C.POINTER(Batch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                      C.c_void_p, C.c_size_t, C.c_void_p]),
    "adkf_fit": (C.c_int, [C.POINTER(Batch), C.c_void_p, C.POINTER(FitOptions), C.c_void_p, C.c_void_p, C.c_void_p,
                           C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "adkf_predict": (C.c_int, [C.POINTER(Batch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                               C.c_void_p, C.c_size_t, C.c_void_p]),
    "adkf_outer_nll_value_grad": (C.c_int, [C.POINTER(Batch), C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                            C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
    "adkf_ift_hypergrad": (C.c_int, [C.POINTER(Batch), C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                     C.c_void_p]),
    "adkf_check_info": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p]),
    "adkf_workspace_bytes_ard": (C.c_size_t, [C.c_int32] * 4),
    "adkf_msg_forward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "adkf_msg_backward_scratch_bytes": (C.c_size_t, [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "adkf_msg_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p,
                                    C.c_size_t, C.c_void_p]),
    "adkf_readout_pool": (C.c_int, [C.c_void_p] * 7 + [C.c_int32] * 5 + [C.c_void_p] * 7),
    "adkf_readout_pool_backward": (C.c_int, [C.c_void_p] * 10 + [C.c_int32] * 5 + [C.c_void_p] * 6),
    "adkf_readout_pool_hidden": (C.c_int, [C.c_void_p] * 4 + [C.c_int32] + [C.c_void_p] * 3 + [C.c_int32] * 5 + [C.c_void_p] * 9),
    "adkf_readout_pool_hidden_backward": (C.c_int, [C.c_void_p] * 2 + [C.c_int32] + [C.c_void_p] * 9 + [C.c_int32] * 5 + [C.c_void_p] * 6),
    "adkf_pna_aggregate": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p,
                                     C.c_void_p]),
    "adkf_pna_aggregate_backward": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                              C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "adkf_pna_aggregate_backward_relu": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                                   C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]),
    "adkf_block_combine": (C.c_int, [C.c_void_p] * 8 + [C.c_float, C.c_int32, C.c_int32] + [C.c_void_p] * 5),
    "adkf_block_combine_scratch_bytes": (C.c_size_t, [C.c_int32, C.c_int32]),
    "adkf_block_combine_backward": (C.c_int, [C.c_void_p] * 11 + [C.c_int32, C.c_int32] + [C.c_void_p] * 7 + [C.c_size_t, C.c_void_p]),
    "adkf_split_planes": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "adkf_split_planes_t": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "adkf_dense_forward": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                     C.c_void_p]),
    "adkf_dense_weight_grad_scratch_bytes": (C.c_size_t, [C.c_int32, C.c_int32, C.c_int32]),
    "adkf_dense_weight_grad": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                         C.c_void_p, C.c_size_t, C.c_void_p]),
    "adkf_grad_sumsq": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p]),
    "adkf_clip_adam_step": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_int32, C.c_float,
                                      C.c_float, C.c_double, C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_void_p]),
    "adkf_clip_adam_step_one": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_float, C.c_float, C.c_double,
                                          C.c_double, C.c_double, C.c_double, C.c_double, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                          C.c_void_p]),
    "adkf_ift_hypergrad_cg": (C.c_int, [C.POINTER(Batch), C.c_void_p, C.c_int32, C.c_int32, C.c_float, C.c_void_p, C.c_void_p,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t,
                                        C.c_void_p]),
    "adkf_double_path_tasks": (C.c_int, [C.POINTER(Batch), C.c_void_p, C.c_void_p, C.c_size_t, C.c_void_p]),
}


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is not built. Run `python -c 'import __graft_entry__ as g; g.build()'` (hipcc, gfx950). "
            "The GP path has no CPU or PyTorch fallback.")
    # torch first: its bundled HIP runtime must be the one in the process before this library binds to libamdhip64
    # (loaded the other way round, the library sees "no ROCm-capable device" for streams created by torch)
    import torch  # noqa: F401

    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError = the library does not export what the header declares
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        detail = f" ({_lib.adkf_last_hip_error().decode()})" if rc == -4 and _lib is not None else ""
        raise RuntimeError(f"{what} failed: {ERRORS.get(rc, rc)}{detail}")
