"""Closed-form (no autograd) float64 numpy restatement of the staged pipeline the HIP kernels run.
TEST INFRASTRUCTURE ONLY (same import rule as gp_oracle.py).

The autograd oracle (gp_oracle.py) states WHAT the reference computes
(fs_mol/models/adaptive_dkt.py:173-191 through fs_mol/utils/cauchy_hypergradient.py:43-161);
this file states the algebra the device code uses to get there without autograd, stage by stage
(stage names match adkf_ift_amd/csrc and DESIGN.md), so a failing GPU parity test can be bisected
by comparing intermediates.  It is itself checked against the autograd oracle's golden vectors in
tests/test_oracle.py.

Notation: phi = (rho_n, rho_s, rho_l) raw; noise = softplus(rho_n)+1e-4, s = softplus(rho_s),
l = softplus(rho_l); u = D2 / l^2; K = s kappa(u); A = K_ss + noise I.
"""
from __future__ import annotations

import math

import numpy as np

NOISE_LB = 1e-4
LOG_2PI = math.log(2 * math.pi)
SQRT5 = math.sqrt(5.0)


def _softplus(x):
    return np.logaddexp(0.0, x)


def _sigmoid(x):
    return 1.0 / (1.0 + np.exp(-x))


def transforms(phi):
    """value, first and second derivative of each transformed hyper-parameter wrt its raw value."""
    sg = _sigmoid(phi)
    val = _softplus(phi)
    val[0] += NOISE_LB
    return val, sg, sg * (1.0 - sg)


def kappa(u, kind):
    """kappa(u), kappa'(u), kappa''(u) with u = squared scaled distance."""
    if kind == 0:
        k = np.exp(-0.5 * u)
        return k, -0.5 * k, 0.25 * k
    r = np.sqrt(u)
    e = np.exp(-SQRT5 * r)
    return (1 + SQRT5 * r + 5.0 / 3.0 * u) * e, -(5.0 / 6.0) * (1 + SQRT5 * r) * e, (25.0 / 12.0) * e


def sqdist(X, Y):
    return ((X[:, None, :] - Y[None, :, :]) ** 2).sum(-1)


def median_lengthscale(D2):
    n = D2.shape[0]
    vals = D2[np.triu_indices(n, 1)]
    vals = np.sort(vals[vals > 0])
    return math.sqrt(0.5 * vals[(len(vals) - 1) // 2])


def lognormal_terms(x, loc, scale):
    """log p(x), d/dx, d2/dx2 of the LogNormal prior."""
    lx = math.log(x)
    lp = -lx - math.log(scale) - 0.5 * LOG_2PI - (lx - loc) ** 2 / (2 * scale ** 2)
    d1 = (-1.0 - (lx - loc) / scale ** 2) / x
    d2 = (1.0 + (lx - loc) / scale ** 2 - 1.0 / scale ** 2) / x ** 2
    return lp, d1, d2


def inner_stage(D2ss, y, phi, pri, kind, want_hessian=True):
    """Stage B/C: f_in, grad_phi f_in, (H), plus the matrices later stages reuse."""
    n = len(y)
    (noise, s, l), d1, d2 = transforms(np.asarray(phi, dtype=np.float64))
    u = D2ss / l ** 2
    k0, k1, k2 = kappa(u, kind)
    K = s * k0
    A = K + noise * np.eye(n)
    L = np.linalg.cholesky(A)
    Ainv = np.linalg.inv(A)
    alpha = Ainv @ y
    logdet = 2 * np.log(np.diag(L)).sum()
    nll = 0.5 * y @ alpha + 0.5 * logdet + 0.5 * n * LOG_2PI
    lpn, dpn, d2pn = lognormal_terms(noise, pri[0], pri[1])
    lpl, dpl, d2pl = (0.0, 0.0, 0.0)
    if pri[3] > 0:
        lpl, dpl, d2pl = lognormal_terms(l, pri[2], pri[3])
    f_in = (nll - lpn - lpl) / n

    G = s * k1 * (-2 * u / l)                     # dK/dl
    trAinv = np.trace(Ainv)
    aa = alpha @ alpha
    trAinvG = (Ainv * G).sum()
    aGa = alpha @ G @ alpha
    # d nll / d(noise, s, l)
    g_noise = 0.5 * trAinv - 0.5 * aa
    g_s = (0.5 * (n - noise * trAinv) - 0.5 * (y @ alpha - noise * aa)) / s
    g_l = 0.5 * trAinvG - 0.5 * aGa
    g_in = np.array([(g_noise - dpn) * d1[0], g_s * d1[1], (g_l - dpl) * d1[2]]) / n
    out = dict(f_in=f_in, g_in=g_in, Ainv=Ainv, alpha=alpha, G=G, K=K, u=u, k1=k1, k2=k2,
               noise=noise, s=s, l=l, d1=d1, d2=d2, logdet=logdet)
    if not want_hessian:
        return out

    P = Ainv @ G
    gamma = Ainv @ alpha
    beta = G @ alpha
    delta = Ainv @ beta
    trA2 = (Ainv * Ainv).sum()
    trPA = (P * Ainv).sum()                      # tr(A^-1 G A^-1)
    trPP = (P * P.T).sum()                       # tr(A^-1 G A^-1 G)
    ag = alpha @ gamma
    bg = beta @ gamma
    bd = beta @ delta
    ab = alpha @ beta
    ya = y @ alpha
    Kll = s * (k2 * 4 * u * u / l ** 2 + k1 * 6 * u / l ** 2)     # d2K/dl2
    trAinvKll = (Ainv * Kll).sum()
    aKlla = alpha @ Kll @ alpha
    # second derivatives of nll wrt the TRANSFORMED parameters (noise, s, l)
    h = np.zeros((3, 3))
    h[0, 0] = ag - 0.5 * trA2
    h[0, 1] = ((aa - noise * ag) - 0.5 * (trAinv - noise * trA2)) / s
    h[0, 2] = bg - 0.5 * trPA
    h[1, 1] = ((ya - 2 * noise * aa + noise ** 2 * ag) - 0.5 * (n - 2 * noise * trAinv + noise ** 2 * trA2)) / s ** 2
    h[1, 2] = ((ab - noise * bg) - 0.5 * (trAinvG - noise * trPA)) / s - (0.5 * aGa - 0.5 * trAinvG) / s
    h[2, 2] = bd - 0.5 * aKlla - 0.5 * trPP + 0.5 * trAinvKll
    h[1, 0], h[2, 0], h[2, 1] = h[0, 1], h[0, 2], h[1, 2]
    gt = np.array([g_noise - dpn, g_s, g_l - dpl])          # d(nll - logpriors)/d transformed
    h[0, 0] -= d2pn
    h[2, 2] -= d2pl
    H = (h * np.outer(d1, d1) + np.diag(gt * d2)) / n
    out.update(H=H, P=P, gamma=gamma, beta=beta, delta=delta)
    return out


def outer_stage(D2qs, D2qq, yq, ys, inner, kind):
    """Stage D: f_out, grad_phi f_out and the cotangent matrices M_A, M_B, Omega."""
    noise, s, l, d1 = inner["noise"], inner["s"], inner["l"], inner["d1"]
    Ainv, alpha = inner["Ainv"], inner["alpha"]
    m = len(yq)
    uqs, uqq = D2qs / l ** 2, D2qq / l ** 2
    kqs0, kqs1, _ = kappa(uqs, kind)
    kqq0, kqq1, _ = kappa(uqq, kind)
    B = s * kqs0
    C = B @ Ainv
    mu = B @ alpha
    S = s * kqq0 - C @ B.T + noise * np.eye(m)
    Sinv = np.linalg.inv(S)
    r = yq - mu
    e = Sinv @ r
    f_out = 0.5 * r @ e + 0.5 * np.linalg.slogdet(S)[1] + 0.5 * m * LOG_2PI
    Om = 0.5 * (Sinv - np.outer(e, e))
    OC = Om @ C
    M_B = -2 * OC - np.outer(e, alpha)
    Cte = C.T @ e
    M_A = C.T @ OC + 0.5 * (np.outer(Cte, alpha) + np.outer(alpha, Cte))
    u = inner["u"]
    k1 = inner["k1"]
    # derivatives wrt transformed params
    g_noise = np.trace(Om) + np.trace(M_A)
    g_s = ((M_A * inner["K"]).sum() + (M_B * B).sum() + (Om * (s * kqq0)).sum()) / s
    g_l = ((M_A * (s * k1 * (-2 * u / l))).sum() + (M_B * (s * kqs1 * (-2 * uqs / l))).sum()
           + (Om * (s * kqq1 * (-2 * uqq / l))).sum())
    g_out = np.array([g_noise, g_s, g_l]) * d1
    # weights on the squared distances: d f_out / d D2
    W_ss = M_A * s * k1 / l ** 2
    W_qs = M_B * s * kqs1 / l ** 2
    W_qq = Om * s * kqq1 / l ** 2
    return dict(f_out=f_out, g_out=g_out, W_ss=W_ss, W_qs=W_qs, W_qq=W_qq, mean=mu, cov=S, M_A=M_A, M_B=M_B, Om=Om)


def mixed_stage(v, inner, n):
    """Stage F: weights on D2_ss of g(Z) = v^T grad_phi f_in(Z, phi)."""
    noise, s, l, d1 = inner["noise"], inner["s"], inner["l"], inner["d1"]
    Ainv, alpha, P, u, k1, k2 = inner["Ainv"], inner["alpha"], inner["P"], inner["u"], inner["k1"], inner["k2"]
    cn, cs, cl = v[0] * d1[0], v[1] * d1[1] / s, v[2] * d1[2]
    # X = A^-1 B_v with B_v = cn I + cs K + cl G
    X = cn * Ainv + cs * (np.eye(n) - noise * Ainv) + cl * P
    w = cn * inner["gamma"] + cs * (alpha - noise * inner["gamma"]) + cl * inner["delta"]
    dg_dA = (-0.5 * X @ Ainv + 0.5 * (np.outer(w, alpha) + np.outer(alpha, w))) / n
    Q = 0.5 * (Ainv - np.outer(alpha, alpha)) / n
    dBv_du = cs * s * k1 + cl * s * (-2.0 / l) * (k1 + u * k2)
    return dg_dA * s * k1 / l ** 2 + Q * dBv_du / l ** 2


def dz_from_weights(Zs, Zq, W_ss, W_qs, W_qq):
    """Stage G: chain d/dD2 -> d/dZ (D2_ij = |z_i - z_j|^2)."""
    Wt = W_ss + W_ss.T
    dZs = 2 * (Wt.sum(1)[:, None] * Zs - Wt @ Zs)
    dZq = np.zeros_like(Zq)
    if W_qs is not None:
        dZs += 2 * (W_qs.sum(0)[:, None] * Zs - W_qs.T @ Zq)
        Wq = W_qq + W_qq.T
        dZq = 2 * (W_qs.sum(1)[:, None] * Zq - W_qs @ Zs) + 2 * (Wq.sum(1)[:, None] * Zq - Wq @ Zq)
    return dZs, dZq


def full_pipeline(Zs, ys, Zq, yq, phi, pri, kind):
    Zs, Zq, ys, yq = (np.asarray(a, dtype=np.float64) for a in (Zs, Zq, ys, yq))
    n = len(ys)
    D2ss, D2qs, D2qq = sqdist(Zs, Zs), sqdist(Zq, Zs), sqdist(Zq, Zq)
    inner = inner_stage(D2ss, ys, phi, pri, kind)
    outer = outer_stage(D2qs, D2qq, yq, ys, inner, kind)
    v = np.linalg.solve(inner["H"], outer["g_out"])
    W_mixed = mixed_stage(v, inner, n)
    W_in = 0.5 * (inner["Ainv"] - np.outer(inner["alpha"], inner["alpha"])) / n * inner["s"] * inner["k1"] / inner["l"] ** 2
    dfin_dZs, _ = dz_from_weights(Zs, Zq, W_in, None, None)
    dZs_dir, dZq_dir = dz_from_weights(Zs, Zq, outer["W_ss"], outer["W_qs"], outer["W_qq"])
    mixed_Zs, _ = dz_from_weights(Zs, Zq, W_mixed, None, None)
    dZs_tot, dZq_tot = dz_from_weights(Zs, Zq, outer["W_ss"] - W_mixed, outer["W_qs"], outer["W_qq"])
    return dict(l0=median_lengthscale(D2ss), f_in=inner["f_in"], g_in=inner["g_in"], H=inner["H"], f_out=outer["f_out"],
                g_out=outer["g_out"], v=v, dfin_dZs=dfin_dZs, dZs_direct=dZs_dir, dZq_direct=dZq_dir,
                mixed_Zs=mixed_Zs, dZs_total=dZs_tot, dZq_total=dZq_tot, pred_mean=outer["mean"],
                pred_var=np.diag(outer["cov"]).copy())
