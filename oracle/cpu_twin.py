"""ctypes/numpy binding of libadkf_gp_cpu.so, the C++ CPU twin of the GP entry points of include/adkf_gp.h
(adkf_ift_amd/csrc/cpu/adkf_gp_cpu.cpp).  TEST / BASELINE INFRASTRUCTURE ONLY (same import rule as gp_oracle.py): used by
tests/test_cpu_twin.py and by bench.py's cpu_baseline leg ("kind": "twin").  The product package never imports this module
and its operators keep refusing CPU tensors."""
from __future__ import annotations

import ctypes as C
import os
import subprocess
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "adkf_ift_amd", "csrc", "cpu", "adkf_gp_cpu.cpp")
LIB = os.path.join(ROOT, "adkf_ift_amd", "libadkf_gp_cpu.so")


class Batch(C.Structure):   # adkf_batch_t
    _fields_ = [("T", C.c_int32), ("ns_max", C.c_int32), ("nq_max", C.c_int32), ("d", C.c_int32), ("kernel", C.c_int32), ("flags", C.c_int32),
                ("n_s", C.c_void_p), ("n_q", C.c_void_p), ("Z_s", C.c_void_p), ("y_s", C.c_void_p), ("Z_q", C.c_void_p), ("y_q", C.c_void_p),
                ("priors", C.c_void_p)]


class FitOptions(C.Structure):   # adkf_fit_options_t
    _fields_ = [("max_evals", C.c_int32), ("exact_evals", C.c_int32), ("gtol", C.c_float), ("ftol", C.c_float), ("ev_start", C.c_void_p),
                ("ev_stop", C.c_void_p)]


def build(force: bool = False) -> str:
    """g++ -O3 -fopenmp (the image's toolchain); rebuilt when the source is newer than the library."""
    if force or not os.path.exists(LIB) or os.path.getmtime(LIB) < os.path.getmtime(SRC):
        subprocess.check_call(["g++", "-O3", "-fopenmp", "-shared", "-fPIC", "-std=c++17", "-I", os.path.join(ROOT, "include"), SRC, "-o", LIB])
    return LIB


_lib = None


def load():
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
    return _lib


def _f32(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


class CpuBatch:
    """Host arrays of one batch + its adkf_batch_t."""

    def __init__(self, Z_s, y_s, priors, kind, Z_q=None, y_q=None, n_s=None, n_q=None):
        self.Z_s, self.y_s, self.Z_q, self.y_q, self.priors = _f32(Z_s), _f32(y_s), _f32(Z_q), _f32(y_q), _f32(priors)
        self.n_s = None if n_s is None else np.ascontiguousarray(n_s, dtype=np.int32)
        self.n_q = None if n_q is None else np.ascontiguousarray(n_q, dtype=np.int32)
        self.T, self.ns, self.d = self.Z_s.shape
        self.nq = 0 if self.Z_q is None else self.Z_q.shape[1]
        self.c = Batch(self.T, self.ns, self.nq, self.d, int(kind), 0, _p(self.n_s), _p(self.n_q), _p(self.Z_s), _p(self.y_s), _p(self.Z_q),
                       _p(self.y_q), _p(self.priors))


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} (CPU twin) returned {rc}")


def init_params(Z_s, numeric=False, use_ls_prior=True, n_s=None):
    Z_s = _f32(Z_s)
    T = Z_s.shape[0]
    b = CpuBatch(Z_s, np.zeros(Z_s.shape[:2], np.float32), np.zeros((T, 4), np.float32), 0, n_s=n_s)
    phi, pri, l0 = np.empty((T, 3), np.float32), np.empty((T, 4), np.float32), np.empty(T, np.float32)
    _check(load().adkf_init_params(C.byref(b.c), int(numeric), int(use_ls_prior), _p(phi), _p(pri), _p(l0), None, C.c_size_t(0), None), "adkf_init_params")
    return phi, pri, l0


def mll_value_grad(b: CpuBatch, phi, want_dZ=True):
    phi = _f32(phi)
    f, g, info = np.empty(b.T, np.float32), np.empty((b.T, 3), np.float32), np.empty(b.T, np.int32)
    dZ = np.empty((b.T, b.ns, b.d), np.float32) if want_dZ else None
    _check(load().adkf_mll_value_grad(C.byref(b.c), _p(phi), _p(f), _p(g), _p(dZ), _p(info), None, C.c_size_t(0), None), "adkf_mll_value_grad")
    return f, g, dZ, info


def fit(b: CpuBatch, phi0, max_evals=200, gtol=1e-5, ftol=2.220446049250313e-09, exact_evals=False):
    phi = _f32(phi0).copy()
    f, gn, ne, info = np.empty(b.T, np.float32), np.empty(b.T, np.float32), np.empty(b.T, np.int32), np.empty(b.T, np.int32)
    opt = FitOptions(int(max_evals), int(exact_evals), float(gtol), float(ftol), None, None)
    _check(load().adkf_fit(C.byref(b.c), _p(phi), C.byref(opt), _p(f), _p(gn), _p(ne), _p(info), None, C.c_size_t(0), None), "adkf_fit")
    return phi, f, gn, ne, info


def predict(b: CpuBatch, phi, want_cov=False):
    phi = _f32(phi)
    mean, var, info = np.empty((b.T, b.nq), np.float32), np.empty((b.T, b.nq), np.float32), np.empty(b.T, np.int32)
    cov = np.empty((b.T, b.nq, b.nq), np.float32) if want_cov else None
    _check(load().adkf_predict(C.byref(b.c), _p(phi), _p(mean), _p(var), _p(cov), _p(info), None, C.c_size_t(0), None), "adkf_predict")
    return mean, var, cov, info


def ift_hypergrad(b: CpuBatch, phi, flags=0):
    phi = _f32(phi)
    out = dict(f_out=np.empty(b.T, np.float32), dZ_s=np.empty((b.T, b.ns, b.d), np.float32), dZ_q=np.empty((b.T, b.nq, b.d), np.float32),
               g_phi=np.empty((b.T, 3), np.float32), v=np.empty((b.T, 3), np.float32), H=np.empty((b.T, 3, 3), np.float32), info=np.empty(b.T, np.int32))
    _check(load().adkf_ift_hypergrad(C.byref(b.c), _p(phi), int(flags), _p(out["f_out"]), _p(out["dZ_s"]), _p(out["dZ_q"]), _p(out["g_phi"]),
                                     _p(out["v"]), _p(out["H"]), _p(out["info"]), None, C.c_size_t(0), None), "adkf_ift_hypergrad")
    return out


def outer_nll_value_grad(b: CpuBatch, phi):
    phi = _f32(phi)
    f, g, info = np.empty(b.T, np.float32), np.empty((b.T, 3), np.float32), np.empty(b.T, np.int32)
    dZs, dZq = np.empty((b.T, b.ns, b.d), np.float32), np.empty((b.T, b.nq, b.d), np.float32)
    _check(load().adkf_outer_nll_value_grad(C.byref(b.c), _p(phi), _p(f), _p(g), _p(dZs), _p(dZq), _p(info), None, C.c_size_t(0), None),
           "adkf_outer_nll_value_grad")
    return f, g, dZs, dZq, info


def time_tasks(tasks, kind: int, inner_evals: int, budget_s: float = 15.0, regression: bool = False):
    """bench.py's second CPU baseline: the unit of work of SURVEY section 8(d) - fresh parameters, exactly `inner_evals` MLL
    value+gradient evaluations, IFT hypergradient - on growing prefixes of the same synthetic workload until `budget_s` is used.
    Returns (tasks per second, tasks done, OpenMP threads)."""
    Zs, Zq = tasks.features()
    Zs, Zq, ys, yq = (a.numpy() for a in (Zs, Zq, tasks.y_s, tasks.y_q))
    load()
    # the cores we actually own, like the "port" baseline (oracle/ref_cpu_path.py): a GPU box shows 256 logical CPUs of which a
    # one-GPU job has about 16
    avail = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    threads = min(avail, 16)
    C.CDLL("libgomp.so.1").omp_set_num_threads(threads)
    done, t_used, chunk = 0, 0.0, min(len(Zs), max(threads, 16))
    while done < len(Zs) and t_used < budget_s:
        sl = slice(done, min(len(Zs), done + chunk))
        t0 = time.perf_counter()
        phi0, pri, _ = init_params(Zs[sl], numeric=regression)
        b = CpuBatch(Zs[sl], ys[sl], pri, kind, Z_q=Zq[sl], y_q=yq[sl])
        phi, *_ = fit(b, phi0, inner_evals, exact_evals=True)
        ift_hypergrad(b, phi)
        t_used += time.perf_counter() - t0
        done += sl.stop - sl.start
    return done / t_used, done, threads
