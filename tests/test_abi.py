"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/adkf_gp.h declares.
No compute call is made here (there is no GPU); host-only entry points are exercised."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    import __graft_entry__ as ge
    from adkf_ift_amd import _lib

    ge.build()
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    from adkf_ift_amd import _lib

    header = open(os.path.join(ROOT, "include", "adkf_gp.h")).read()
    declared = set(re.findall(r"\b(adkf_[a-z_]+)\s*\(", header))
    assert len(declared) >= 11
    assert declared == set(_lib.SIGNATURES), declared ^ set(_lib.SIGNATURES)
    for name in declared:
        assert hasattr(lib, name), name


def test_host_only_entry_points(lib):
    assert b"gfx950" in lib.adkf_version()
    assert lib.adkf_max_points() >= 256
    assert lib.adkf_workspace_bytes(0, 8, 8, 4) == 0
    small = lib.adkf_workspace_bytes(4, 16, 16, 8)
    big = lib.adkf_workspace_bytes(256, 128, 128, 256)
    assert 0 < small < big
    assert big > 11 * 256 * 128 * 128 * 4  # the eleven N x N work matrices


def test_bad_arguments_are_rejected_without_touching_the_gpu(lib):
    import ctypes as C

    from adkf_ift_amd._lib import Batch

    b = Batch()
    b.T, b.ns_max, b.nq_max, b.d, b.kernel = 0, 8, 8, 4, 0
    assert lib.adkf_median_lengthscale(C.byref(b), None, None, 0, None) == -1
    b.T, b.Z_s = 2, 1  # non-null dummy pointer; rejected before any launch
    b.ns_max = 100000
    assert lib.adkf_median_lengthscale(C.byref(b), None, None, 0, None) == -2
    b.ns_max, b.kernel = 8, 7
    assert lib.adkf_median_lengthscale(C.byref(b), None, None, 0, None) == -1


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from adkf_ift_amd import _lib

    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(RuntimeError, match="no CPU or PyTorch fallback"):
        _lib.load()


def test_cpu_tensors_are_refused():
    import torch

    from adkf_ift_amd import gp_ops

    with pytest.raises(RuntimeError, match="no CPU fallback"):
        gp_ops.GPBatch(torch.zeros(1, 4, 2), torch.zeros(1, 4), torch.zeros(1, 4))


def test_ctypes_signatures_agree_with_the_header(lib):
    """Every ctypes prototype in adkf_ift_amd/_lib.py has the argument count AND the scalar classes (pointer / integer /
    float / double) of the C declaration - a float passed where the header says double would be silently misread."""
    import ctypes as C

    from adkf_ift_amd import _lib

    header = open(os.path.join(ROOT, "include", "adkf_gp.h")).read()
    header = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    protos = dict(re.findall(r"\b(adkf_[a-z_]+)\s*\(([^)]*)\)\s*;", header))
    assert set(protos) == set(_lib.SIGNATURES)

    def c_class(decl):
        decl = decl.strip()
        if decl in ("void", ""):
            return None
        if "*" in decl:
            return "ptr"
        ty = decl.rsplit(" ", 1)[0].replace("const", "").strip()
        return {"int32_t": "i32", "int": "i32", "int64_t": "i64", "size_t": "size", "float": "f32", "double": "f64"}[ty]

    def py_class(t):
        if t in (C.c_void_p, C.c_char_p) or (isinstance(t, type) and issubclass(t, C._Pointer)):
            return "ptr"
        return {C.c_int32: "i32", C.c_int: "i32", C.c_int64: "i64", C.c_size_t: "size", C.c_float: "f32", C.c_double: "f64"}[t]

    for name, args in protos.items():
        want = [c for c in (c_class(a) for a in args.split(",")) if c is not None]
        got = [py_class(t) for t in _lib.SIGNATURES[name][1]]
        assert want == got, (name, want, got)


def test_ctypes_structs_mirror_the_header():
    """adkf_batch_t / adkf_fit_options_t: same field names, order and scalar classes as the ctypes.Structure mirrors."""
    import ctypes as C

    from adkf_ift_amd import _lib

    header = open(os.path.join(ROOT, "include", "adkf_gp.h")).read()
    header = re.sub(r"/\*.*?\*/", " ", header, flags=re.S)
    for cname, mirror in (("adkf_batch", _lib.Batch), ("adkf_fit_options", _lib.FitOptions)):
        body = re.search(r"typedef struct %s\s*\{(.*?)\}" % cname, header, flags=re.S).group(1)
        fields = []
        for decl in (d.strip() for d in body.split(";")):
            if not decl:
                continue
            name = re.findall(r"[A-Za-z_0-9]+", decl)[-1]
            kind = "ptr" if "*" in decl else {"int32_t": "i32", "float": "f32"}[decl.split()[0]]
            fields.append((name, kind))
        got = [(n, "ptr" if t is C.c_void_p else {C.c_int32: "i32", C.c_float: "f32"}[t]) for n, t in mirror._fields_]
        assert fields == got, (cname, fields, got)


@pytest.mark.parametrize("ns,nq", [(32, 32), (128, 128), (200, 256)])
def test_entry_points_fail_cleanly_without_a_device(lib, ns, nq):
    """Host side of the entry points on a box WITHOUT a GPU: argument checks, workspace carving and the launch sequence run, every
    launch is refused by the runtime and the call returns ADKF_E_LAUNCH - it must not crash (round 5: a host-side recursion in the
    workspace helper took the whole process down on the first call, and no CPU test went through an entry point).  Host memory stands
    in for device memory: nothing is ever dereferenced on the host."""
    import ctypes as C

    import torch
    if torch.cuda.is_available():
        pytest.skip("this is the no-device check")
    from adkf_ift_amd import _lib

    T, d = 3, 8
    nb = lib.adkf_workspace_bytes(T, ns, nq, d)
    ws = torch.zeros(nb // 4 + 64)
    Zs, Zq, ys, yq = torch.zeros(T, ns, d), torch.zeros(T, nq, d), torch.zeros(T, ns), torch.zeros(T, nq)
    pri, phi, l0, f, g, info = torch.zeros(T, 4), torch.zeros(T, 3), torch.zeros(T), torch.zeros(T), torch.zeros(T, 3), torch.zeros(T, dtype=torch.int32)
    b = _lib.Batch()
    b.T, b.ns_max, b.nq_max, b.d, b.kernel, b.flags = T, ns, nq, d, 0, 0
    b.n_s = b.n_q = None
    b.Z_s, b.y_s, b.Z_q, b.y_q, b.priors = Zs.data_ptr(), ys.data_ptr(), Zq.data_ptr(), yq.data_ptr(), pri.data_ptr()
    p = lambda t: C.c_void_p(t.data_ptr())
    for flags in (0, 16, 32):   # ADKF_BATCH_LG_UNFUSED, ADKF_BATCH_LG_FUSED
        b.flags = flags
        assert lib.adkf_init_params(C.byref(b), 0, 1, p(phi), p(pri), p(l0), p(ws), nb, None) == -4
        assert lib.adkf_mll_value_grad(C.byref(b), p(phi), p(f), p(g), None, p(info), p(ws), nb, None) == -4
        opt = _lib.FitOptions(5, 1, 1e-5, 2.22e-9, None, None)
        assert lib.adkf_fit(C.byref(b), p(phi), C.byref(opt), p(f), p(l0), None, p(info), p(ws), nb, None) == -4
    assert b"device" in lib.adkf_last_hip_error() or lib.adkf_last_hip_error()
