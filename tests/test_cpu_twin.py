"""CPU: libadkf_gp_cpu.so - the C++ twin of the seven GP entry points of include/adkf_gp.h -
exports them with the header's signatures and reproduces the golden vectors (float64 autograd oracle) through the same call
sequence the GPU parity test uses.  It is float64 inside, so it is held to 1e-5 (phi and every output cross the boundary as float32)."""
import glob
import os

import numpy as np
import pytest

from oracle import cpu_twin as TW

TOL = 1e-5


def rel(a, ref):
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30)


def _batch(g, pad_s=0, pad_q=0):
    def pad(x, p):
        x = np.asarray(x, dtype=np.float32)
        return np.concatenate([x, np.full((p, *x.shape[1:]), 7.5, np.float32)])[None] if p else x[None]
    n, m = g["Z_s"].shape[0], g["Z_q"].shape[0]
    b = TW.CpuBatch(pad(g["Z_s"], pad_s), pad(g["y_s"], pad_s), np.asarray(g["priors"], np.float32)[None], int(g["kind"]), Z_q=pad(g["Z_q"], pad_q),
                    y_q=pad(g["y_q"], pad_q), n_s=[n] if pad_s else None, n_q=[m] if pad_q else None)
    return b, np.asarray(g["phi"], np.float32)[None], n, m


def test_exports_every_gp_entry_point_of_the_header():
    lib = TW.load()
    for name in ("adkf_version", "adkf_max_points", "adkf_workspace_bytes", "adkf_median_lengthscale", "adkf_init_params", "adkf_mll_value_grad",
                 "adkf_fit", "adkf_predict", "adkf_outer_nll_value_grad", "adkf_ift_hypergrad"):
        assert hasattr(lib, name), name


def test_golden_cases(golden_dir):
    files = sorted(glob.glob(os.path.join(golden_dir, "gp_*.npz")))
    assert len(files) >= 30
    worst = {}
    for f in files:
        g = np.load(f)
        for pad_s, pad_q in ((0, 0), (3, 5)):
            b, phi, n, m = _batch(g, pad_s, pad_q)
            fin, gin, dZin, info = TW.mll_value_grad(b, phi)
            assert int(info[0]) == 0
            out = TW.ift_hypergrad(b, phi)
            assert int(out["info"][0]) == 0
            mean, var, cov, info = TW.predict(b, phi, want_cov=True)
            fo, go, dzs, dzq, info = TW.outer_nll_value_grad(b, phi)
            _, _, l0 = TW.init_params(b.Z_s, n_s=b.n_s)
            got = {"l0": l0[0], "f_in": fin[0], "g_in": gin[0], "dfin_dZs": dZin[0, :n], "H": out["H"][0], "f_out": out["f_out"][0],
                   "g_out": out["g_phi"][0], "v": out["v"][0], "dZs_total": out["dZ_s"][0, :n], "dZq_total": out["dZ_q"][0, :m],
                   "pred_mean": mean[0, :m], "pred_var": var[0, :m], "dZs_direct": dzs[0, :n], "dZq_direct": dzq[0, :m]}
            if "pred_cov" in g.files:
                got["pred_cov"] = cov[0, :m, :m]
            assert abs(float(fo[0]) - float(g["f_out"])) <= TOL * abs(float(g["f_out"]))
            for k, v in got.items():
                if k not in g.files:      # (the 128-point fixtures store the totals only)
                    continue
                e = rel(v, g[k])
                if k == "g_in" and int(g["fitted"]):
                    e = np.abs(np.asarray(v, dtype=np.float64) - g[k]).max() / 1e-1
                worst[k] = max(worst.get(k, 0.0), e)
                assert e <= TOL, (os.path.basename(f), pad_s, k, e)
            if pad_s:
                assert float(np.abs(out["dZ_s"][0, n:]).max()) == 0.0 and float(np.abs(out["dZ_q"][0, m:]).max()) == 0.0
    print("CPU twin, worst relative errors:", {k: float("%.1e" % v) for k, v in worst.items()})


def test_fit_reaches_the_oracle_optimum(golden_dir):
    """The twin's inner fit (the state machine of csrc/inner.h in float64) against SciPy L-BFGS-B on the float64 oracle."""
    import torch
    from oracle import gp_oracle as O

    for name in ("gp_N32_Nq32_d64_k0_r0_s0", "gp_N16_Nq32_d16_k1_r1_s1", "gp_N64_Nq128_d96_k0_r0_s4"):
        g = np.load(os.path.join(golden_dir, name + ".npz"))
        Zs, ys = torch.tensor(g["Z_s"]), torch.tensor(g["y_s"])
        numeric = "_r1_" in name
        p0, pri = O.init_phi(Zs.double(), numeric, True)
        phi0, pri_tw, _ = TW.init_params(g["Z_s"][None], numeric=numeric)
        assert np.abs(phi0[0] - p0.numpy()).max() <= 1e-5 and rel(pri_tw[0], np.array([pri.noise_loc, pri.noise_scale, pri.ls_loc, pri.ls_scale])) <= 1e-6
        b = TW.CpuBatch(g["Z_s"][None], g["y_s"][None], pri_tw, int(g["kind"]))
        phi, f, gn, ne, info = TW.fit(b, phi0, 300)
        f_star = float(O.f_inner(Zs.double(), ys.double(), O.fit_phi(Zs.double(), ys.double(), p0, pri, int(g["kind"]))[0], pri, int(g["kind"])))
        assert int(info[0]) == 0 and float(f[0]) <= f_star + 1e-6 * abs(f_star) + 1e-7, (name, float(f[0]), f_star)
        assert float(gn[0]) <= 2e-4 and 3 <= int(ne[0]) <= 300


def test_flags_of_the_hypergradient(golden_dir):
    g = np.load(os.path.join(golden_dir, "gp_N16_Nq32_d16_k0_r0_s0.npz"))
    b, phi, n, m = _batch(g)
    o1 = TW.ift_hypergrad(b, phi, flags=1)     # ADKF_IGNORE_GRAD_CORRECTION: the direct part only
    assert rel(o1["dZ_s"][0], g["dZs_direct"]) <= TOL and rel(o1["dZ_q"][0], g["dZq_direct"]) <= TOL
    o2 = TW.ift_hypergrad(b, phi, flags=2)     # ADKF_IGNORE_DIRECT_GRAD: minus the mixed term only
    assert rel(o2["dZ_s"][0], -g["mixed_Zs"]) <= TOL and float(np.abs(o2["dZ_q"][0]).max()) == 0.0
