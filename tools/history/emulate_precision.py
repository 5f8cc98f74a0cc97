#!/usr/bin/env python
"""CPU emulation of the device pipeline's PRECISION choices (design tool for DESIGN.md section 4, not product code).

Runs oracle/closed_form.py's staged algebra with a chosen precision per stage over the random shapes of
tests/test_gpu_stress.py and prints, per configuration, the worst error / tolerance ratio the stress test would see
(tolerance = max(1e-4 * slack, 4 * e32), e32 = the float32 autograd restatement).  Usage:
    python tools/emulate_precision.py [n_cases] [config ...]
Configurations (see CONFIGS): which of {A^-1/alpha, C, S, S^-1/e/f_out, Hessian pieces} are computed in float64 before being
rounded to float32 for the remaining float32 stages."""
import math
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import closed_form as C  # noqa: E402
from oracle import gp_oracle as O  # noqa: E402
from adkf_ift_amd.synthetic import make_tasks  # noqa: E402

f32, f64 = np.float32, np.float64


def d2_gemm_form(X, Y, mu):
    """squared distances as the device builds them: centred float32 rows, |x|^2 + |y|^2 - 2 x.y, clamped, exact-0 diagonal if X is Y"""
    Xc, Yc = (X - mu).astype(f32), (Y - mu).astype(f32)
    nx, ny = (Xc * Xc).sum(1, dtype=f32), (Yc * Yc).sum(1, dtype=f32)
    D = np.maximum(nx[:, None] + ny[None, :] - f32(2) * (Xc @ Yc.T), f32(0))
    if X is Y:
        np.fill_diagonal(D, 0)
        D = np.minimum(D, D.T)
    return D


def sweep_inverse(A):
    """-(sweep of all indices) = A^-1 the way factor.h does it: Gauss-Jordan without pivoting, in A's dtype (unblocked:
    the 4-pivot blocking of the device changes the order of a few operations, not the error class)."""
    M = A.copy()
    n = M.shape[0]
    one = M.dtype.type(1)
    for k in range(n):
        d = M[k, k]
        r = M[k, :] / d
        c = M[:, k].copy()
        M -= np.outer(c, r)
        M[k, :] = r
        M[:, k] = r
        M[k, k] = -one / d
    return -M


def refine_inverse(S, X, steps=1):
    """Mixed-precision refinement of an explicit inverse: residual R = I - S X accumulated in float64 from the float32
    matrices, correction X += X R applied in float32."""
    for _ in range(steps):
        R = (np.eye(S.shape[0]) - S.astype(f64) @ X.astype(f64)).astype(f32)
        X = X + X @ R
    return X


def pipeline(Zs, ys, Zq, yq, phi, pri, kind, cfg):
    """cfg: set of stage names computed in float64 (inputs are always the float32 D2 the device has)."""
    Zs32, Zq32 = Zs.astype(f32), Zq.astype(f32)
    mu = Zs32.mean(0, dtype=f32)
    D2ss, D2qs, D2qq = d2_gemm_form(Zs32, Zs32, mu), d2_gemm_form(Zq32, Zs32, mu), d2_gemm_form(Zq32, Zq32, mu)
    n, m = len(ys), len(yq)
    phi32 = np.asarray(phi, dtype=f32)
    (noise, s, l), d1, d2 = C.transforms(phi32.astype(f64))
    T = lambda name: f64 if name in cfg else f32          # dtype of a stage
    cast = lambda x, name: np.asarray(x, dtype=T(name))

    def kap(D2, name):
        u = cast(D2, name) / cast(l, name) ** 2
        return (u,) + tuple(C.kappa(u, kind))

    # ---- inner: A^-1, alpha (stage "ainv")
    dt = T("ainv")
    u, k0, k1, k2 = kap(D2ss, "ainv")
    K = dt(s) * k0
    A = K + dt(noise) * np.eye(n, dtype=dt)
    Ainv = sweep_inverse(A) if "sweep" in cfg else np.linalg.inv(A)
    if "refine_a" in cfg:
        Ainv = refine_inverse(A, Ainv)
    Ainv = 0.5 * (Ainv + Ainv.T)
    alpha = (Ainv @ ys.astype(dt)) if "alpha_solve" not in cfg else np.linalg.solve(A.astype(f64), ys.astype(f64))
    # what the later float32 stages see
    Ainv32, alpha32 = Ainv.astype(f32), alpha.astype(f32)
    # ---- Hessian pieces (stage "hess"): all traces / mat-vecs
    dt = T("hess")
    uH, k0H, k1H, k2H = kap(D2ss, "hess")
    AiH, alH = Ainv.astype(dt) if "ainv" in cfg and dt == f64 else Ainv32.astype(dt), alpha.astype(dt) if dt == f64 and ("ainv" in cfg or "alpha_solve" in cfg) else alpha32.astype(dt)
    G = dt(s) * k1H * (-2 * uH / dt(l))
    P = AiH @ G
    gamma, beta = AiH @ alH, G @ alH
    delta = AiH @ beta
    y_ = ys.astype(dt)
    trAinv, aa, trAinvG, aGa = np.trace(AiH), alH @ alH, (AiH * G).sum(), alH @ G @ alH
    trA2, trPA, trPP = (AiH * AiH).sum(), (P * AiH).sum(), (P * P.T).sum()
    ag, bg, bd, ab, ya = alH @ gamma, beta @ gamma, beta @ delta, alH @ beta, y_ @ alH
    Kll = dt(s) * (k2H * 4 * uH * uH / dt(l) ** 2 + k1H * 6 * uH / dt(l) ** 2)
    trAinvKll, aKlla = (AiH * Kll).sum(), alH @ Kll @ alH
    lpn, dpn, d2pn = C.lognormal_terms(noise, pri[0], pri[1])
    dpl = d2pl = 0.0
    if pri[3] > 0:
        _, dpl, d2pl = C.lognormal_terms(l, pri[2], pri[3])
    g_noise = 0.5 * trAinv - 0.5 * aa
    g_s = (0.5 * (n - noise * trAinv) - 0.5 * (ya - noise * aa)) / s
    g_l = 0.5 * trAinvG - 0.5 * aGa
    h = np.zeros((3, 3), dtype=dt)
    h[0, 0] = ag - 0.5 * trA2
    h[0, 1] = ((aa - noise * ag) - 0.5 * (trAinv - noise * trA2)) / s
    h[0, 2] = bg - 0.5 * trPA
    h[1, 1] = ((ya - 2 * noise * aa + noise ** 2 * ag) - 0.5 * (n - 2 * noise * trAinv + noise ** 2 * trA2)) / s ** 2
    h[1, 2] = ((ab - noise * bg) - 0.5 * (trAinvG - noise * trPA)) / s - (0.5 * aGa - 0.5 * trAinvG) / s
    h[2, 2] = bd - 0.5 * aKlla - 0.5 * trPP + 0.5 * trAinvKll
    h[1, 0], h[2, 0], h[2, 1] = h[0, 1], h[0, 2], h[1, 2]
    gt = np.array([g_noise - dpn, g_s, g_l - dpl])
    h[0, 0] -= d2pn
    h[2, 2] -= d2pl
    H = ((h * np.outer(d1, d1) + np.diag(gt * d2)) / n).astype(dt)

    # ---- outer: C (stage "c"), S and its inverse / e / f_out (stage "s")
    dt = T("c")
    uqs, kqs0, kqs1, _ = kap(D2qs, "c")
    B = dt(s) * kqs0
    if "c_solve" in cfg:
        Cm = np.linalg.solve(A.astype(f64), B.astype(f64).T).T
    elif "c_refine32" in cfg:
        # one step of iterative refinement in WORKING precision with the explicit inverse as the solver (Skeel: backward stable)
        A32, Ai32_, B32 = A.astype(f32), Ainv32, B.astype(f32)
        C0 = B32 @ Ai32_
        for _ in range(int("c_refine32x2" in cfg) + 1):
            R = B32 - C0 @ A32
            C0 = C0 + R @ Ai32_
        Cm = C0
        a0 = Ai32_ @ ys.astype(f32)
        for _ in range(int("c_refine32x2" in cfg) + 1):
            ra = ys.astype(f32) - A32 @ a0
            a0 = a0 + Ai32_ @ ra
        alpha = a0
        alpha32 = a0
    else:
        Cm = B @ (Ainv.astype(dt) if ("ainv" in cfg and dt == f64) else Ainv32.astype(dt))
    al_c = alpha.astype(dt) if dt == f64 and ("ainv" in cfg or "alpha_solve" in cfg) else alpha32.astype(dt)
    mu_q = B @ al_c
    dt = T("s")
    uqq, kqq0, kqq1, _ = kap(D2qq, "s")
    Cs, Bs = (Cm if "c" in cfg or "c_solve" in cfg else Cm.astype(f32)).astype(dt), (dt(s) * C.kappa(cast(D2qs, "s") / dt(l) ** 2, kind)[0])
    S = dt(s) * kqq0 - Cs @ Bs.T + dt(noise) * np.eye(m, dtype=dt)
    S = 0.5 * (S + S.T)
    Sinv = sweep_inverse(S) if "sweep" in cfg else np.linalg.inv(S)
    if "refine_s" in cfg:
        Sinv = refine_inverse(S, Sinv)
    Sinv = 0.5 * (Sinv + Sinv.T)
    r = yq.astype(dt) - mu_q.astype(dt)
    e = np.linalg.solve(S.astype(f64), r.astype(f64)).astype(dt) if dt == f64 else Sinv @ r
    f_out = float(0.5 * r @ e + 0.5 * np.linalg.slogdet(S.astype(f64) if dt == f64 else S)[1] + 0.5 * m * C.LOG_2PI)
    # ---- everything downstream in float32 from the rounded C, Sinv, e, alpha
    C32, Sinv32, e32v, al32 = Cm.astype(f32), Sinv.astype(f32), e.astype(f32), alpha.astype(f32) if ("ainv" in cfg or "alpha_solve" in cfg) else alpha32
    dw = T("w")
    Cw, Sw, ew, aw = C32.astype(dw), Sinv32.astype(dw), e32v.astype(dw), al32.astype(dw)
    Om = 0.5 * (Sw - np.outer(ew, ew))
    OC = Om @ Cw
    M_B = -2 * OC - np.outer(ew, aw)
    Cte = Cw.T @ ew
    M_A = Cw.T @ OC + 0.5 * (np.outer(Cte, aw) + np.outer(aw, Cte))
    uw, k0w, k1w, k2w = kap(D2ss, "w")
    uqsw, kqs0w, kqs1w, _ = kap(D2qs, "w")
    uqqw, kqq0w, kqq1w, _ = kap(D2qq, "w")
    sw, lw = dw(s), dw(l)
    g_noise_o = np.trace(Om) + np.trace(M_A)
    g_s_o = ((M_A * (sw * k0w)).sum() + (M_B * (sw * kqs0w)).sum() + (Om * (sw * kqq0w)).sum()) / sw
    g_l_o = ((M_A * (sw * k1w * (-2 * uw / lw))).sum() + (M_B * (sw * kqs1w * (-2 * uqsw / lw))).sum()
             + (Om * (sw * kqq1w * (-2 * uqqw / lw))).sum())
    g_out = (np.array([g_noise_o, g_s_o, g_l_o]) * d1).astype(dw)
    W_ss = M_A * sw * k1w / lw ** 2
    W_qs = M_B * sw * kqs1w / lw ** 2
    W_qq = Om * sw * kqq1w / lw ** 2
    v = np.linalg.solve(H.astype(f64), g_out.astype(f64))
    # mixed term (float32 unless "w" is float64)
    cn, cs, cl = v[0] * d1[0], v[1] * d1[1] / s, v[2] * d1[2]
    Aim = Ainv.astype(dw) if ("ainv" in cfg and dw == f64) else Ainv32.astype(dw)
    Pm = (Aim @ (sw * k1w * (-2 * uw / lw)))
    gam_m, del_m = Aim @ aw, Aim @ ((sw * k1w * (-2 * uw / lw)) @ aw)
    X = dw(cn) * Aim + dw(cs) * (np.eye(n, dtype=dw) - dw(noise) * Aim) + dw(cl) * Pm
    wv = dw(cn) * gam_m + dw(cs) * (aw - dw(noise) * gam_m) + dw(cl) * del_m
    dg_dA = (-0.5 * X @ Aim + 0.5 * (np.outer(wv, aw) + np.outer(aw, wv))) / n
    Q = 0.5 * (Aim - np.outer(aw, aw)) / n
    dBv_du = dw(cs) * sw * k1w + dw(cl) * sw * (-2.0 / lw) * (k1w + uw * k2w)
    W_mixed = dg_dA * sw * k1w / lw ** 2 + Q * dBv_du / lw ** 2
    dz = T("dz")
    Zsc, Zqc = (Zs32 - mu).astype(dz), (Zq32 - mu).astype(dz)     # translation invariant: the device uses the raw rows; centred is kinder
    dZs, dZq = C.dz_from_weights(Zsc, Zqc, (W_ss - W_mixed).astype(dz), W_qs.astype(dz), W_qq.astype(dz))
    return dict(f_out=f_out, g_out=g_out.astype(f64), v=v, H=H.astype(f64), dZs_total=dZs.astype(f64), dZq_total=dZq.astype(f64),
                pred_mean=mu_q.astype(f64))


CONFIGS = {
    "fp32": set(),                                              # explicit float32 inverses everywhere (round-1 before ldl.h)
    "ldl": {"alpha_solve", "c_solve"},                          # round 1: C and alpha by a stable solve, the rest float32
    "refC32+sweep": {"c_refine32", "sweep"},                     # C and alpha by ONE float32 refinement step on top of the explicit inverse
    "refC32x2+sweep": {"c_refine32", "c_refine32x2", "sweep"},
    "ldl+sweep": {"alpha_solve", "c_solve", "sweep"},           # the device as it is: C / alpha solved stably, S and A inverted by the sweep
    "ldl+sweep+refS": {"alpha_solve", "c_solve", "sweep", "refine_s"},
    "ldl+sweep+refSA": {"alpha_solve", "c_solve", "sweep", "refine_s", "refine_a"},
    "sweep+refSA": {"sweep", "refine_s", "refine_a"},           # no LDL at all: refined explicit inverses only
    "outer64": {"alpha_solve", "c_solve", "c", "s"},            # + S, S^-1, e, f_out in float64 (rounded to float32 afterwards)
    "outer64+ainv": {"ainv", "c", "s"},                         # + A^-1 in float64 (rounded) for the Hessian / mixed term
    "outer64+ainv+hess": {"ainv", "c", "s", "hess"},            # + Hessian traces in float64
    "all64w": {"ainv", "c", "s", "hess", "w"},                  # + cotangent algebra in float64 (only D2 and the dZ GEMMs float32)
    "all64": {"ainv", "c", "s", "hess", "w", "dz"},
}


def _rel(a, ref):
    a, ref = np.asarray(a, dtype=f64), np.asarray(ref, dtype=f64)
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30)


def main():
    from test_gpu_stress import _random_case
    ncases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
    names = sys.argv[2:] or list(CONFIGS)
    rng = np.random.default_rng(20260)
    worst = {c: {} for c in names}
    fails = {c: [] for c in names}
    for case in range(ncases):
        N, Nq, d, kind, regression, n_s, n_q = _random_case(rng)
        tasks = make_tasks(3, N, d, N_q=Nq, regression=regression, first_task=100 * case)
        Zs, Zq = tasks.features()
        for t in range(3):
            n, m = n_s[t], n_q[t]
            zs, zq, ys, yq = Zs[t, :n], Zq[t, :m], tasks.y_s[t, :n], tasks.y_q[t, :m]
            p0, opri = O.init_phi(zs.double(), regression, True)
            phi = O.fit_phi(zs.double(), ys.double(), p0, opri, kind)[0]
            phi = phi.float().double()                        # the device holds phi in float32
            q = O.full_reference_quantities(zs, ys, zq, yq, phi, opri, kind)
            O.DT = torch.float32
            try:
                q32 = O.full_reference_quantities(zs, ys, zq, yq, phi.float(), opri, kind)
            finally:
                O.DT = torch.float64
            g_floor = 1e-2 * abs(q["f_out"])
            ld_q = float(np.linalg.slogdet(q["pred_cov"])[1])
            quad = 2.0 * q["f_out"] - ld_q - m * math.log(2.0 * math.pi)
            terms = 0.5 * (abs(quad) + abs(ld_q) + m * math.log(2.0 * math.pi))
            slack = {"f_out": max(1.0, terms / abs(q["f_out"])), "g_out": max(1.0, g_floor / np.abs(q["g_out"]).max()),
                     "v": max(1.0, np.abs(np.linalg.inv(q["H"])).sum(1).max() * max(g_floor, np.abs(q["g_out"]).max()) / np.abs(q["v"]).max())}
            noise, os_, ls = O.transform_phi(phi)
            A = O.kernel_matrix(zs.double(), zs.double(), os_, ls, kind) + noise * torch.eye(n, dtype=torch.float64)
            condA, condS, condH = float(torch.linalg.cond(A)), float(np.linalg.cond(q["pred_cov"])), float(np.linalg.cond(q["H"]))
            for c in names:
                out = pipeline(zs.numpy().astype(f64), ys.numpy().astype(f64), zq.numpy().astype(f64), yq.numpy().astype(f64),
                               phi.numpy(), opri.as_array(), kind, CONFIGS[c])
                for k in ("f_out", "g_out", "v", "H", "dZs_total", "dZq_total", "pred_mean"):
                    e, e32 = _rel(out[k], q[k]), _rel(q32[k], q[k])
                    tol = max(1e-4 * slack.get(k, 1.0), 4.0 * e32)
                    ratio = e / tol
                    if ratio > worst[c].get(k, (0,))[0]:
                        worst[c][k] = (ratio, e, e32, case, t, condA, condS, condH)
                    if ratio > 1.0:
                        fails[c].append((case, t, k, float("%.2e" % e), float("%.2e" % e32), float("%.1e" % condA), float("%.1e" % condS), float("%.1e" % condH)))
        if case % 10 == 9:
            print("case", case, "done", flush=True)
    for c in names:
        print("==", c, "fails:", len(fails[c]))
        for k, w in worst[c].items():
            print("   %-10s worst err/tol %6.2f  (err %.2e, fp32-autograd %.2e, case %d task %d, condA %.1e condS %.1e condH %.1e)" % ((k,) + w))
        for f in fails[c][:12]:
            print("   FAIL", f)


if __name__ == "__main__":
    main()
