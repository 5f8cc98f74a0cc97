"""Closed forms for the ARD kernel (one lengthscale per feature dimension; fs_mol/models/adaptive_dkt.py:107-108,
fs_mol/utils/gp_utils.py:27-30 ``ard_num_dims``).  TEST INFRASTRUCTURE ONLY (see gp_oracle.py for the parity status).

Everything reduces to the non-ARD forms of closed_form.py evaluated on SCALED features  z~ = (z - mean_s) / l  at unit
lengthscale, because the kernel depends on z_k / l_k only:

    d f / d l_k  = -(1 / l_k) sum_i z~_ik  d f / d z~_ik                (Euler homogeneity; f translation invariant)
    d f / d z_ik =  (1 / l_k) d f / d z~_ik

and a Hessian-vector product in (noise, outputscale, l_1..l_d) is a directional derivative of (dF/dnoise, dF/ds,
dF/dZ~) along (u_n, u_s, Zdot = Z~ * c),  c_k = -u_lk / l_k  - two O(N^3) products and two O(N^2 d) products, never
an h x h matrix.  ``hvp`` below is what the HIP path implements for its conjugate-gradient solve; the dense ``hessian``
(h calls of hvp) exists for the tests only.
"""
from __future__ import annotations

import math

import numpy as np

from .closed_form import (LOG_2PI, NOISE_LB, dz_from_weights, inner_stage, kappa, lognormal_terms, outer_stage, sqdist,
                          transforms)

RAW_ONE = math.log(math.expm1(1.0))  # softplus(RAW_ONE) = 1


def _weighted_sqdist(Z, w):
    """sum_k w_k (z_ik - z_jk)^2"""
    wn = (Z * Z * w).sum(1)
    return wn[:, None] + wn[None, :] - 2.0 * (Z * w) @ Z.T


class ArdTask:
    """All per-task state at hyper-parameters phi = (raw_noise, raw_outputscale, raw_l_1..d)."""

    def __init__(self, Zs, ys, phi, pri, kind, Zq=None, yq=None):
        self.Zs, self.ys = np.asarray(Zs, np.float64), np.asarray(ys, np.float64)
        self.phi = np.asarray(phi, np.float64)
        self.pri, self.kind = pri, kind
        self.n, self.d = self.Zs.shape
        (val, self.d1, self.d2) = transforms(self.phi.copy())
        self.noise, self.s, self.ell = val[0], val[1], val[2:]
        self.mu = self.Zs.mean(0)
        self.Zt = (self.Zs - self.mu) / self.ell
        self.D2 = sqdist(self.Zt, self.Zt)
        phi3 = np.array([self.phi[0], self.phi[1], RAW_ONE])
        self.inner = inner_stage(self.D2, self.ys, phi3, [pri[0], pri[1], 0.0, -1.0], kind, want_hessian=False)
        self.k0, self.k1, self.k2 = kappa(self.D2, kind)
        Ainv, alpha = self.inner["Ainv"], self.inner["alpha"]
        self.Q = 0.5 * (Ainv - np.outer(alpha, alpha))
        self.W = self.Q * self.s * self.k1 / self.n
        self.G, _ = dz_from_weights(self.Zt, None, self.W, None, None)       # d f_in / d Z~
        self.S1 = (self.Zt * self.G).sum(0)
        lp = d1p = d2p = np.zeros(self.d)
        if pri[3] > 0:
            t = [lognormal_terms(l, pri[2], pri[3]) for l in self.ell]
            lp, d1p, d2p = (np.array([x[q] for x in t]) for q in range(3))
        self.d2pl = d2p
        _, _, self.d2pn = lognormal_terms(self.noise, pri[0], pri[1])
        self.f_in = self.inner["f_in"] - lp.sum() / self.n
        gt_l = -self.S1 / self.ell - d1p / self.n
        self.gt = np.concatenate([self.inner["g_in"][:2] / self.d1[:2], gt_l])   # d f_in / d transformed
        self.g_in = self.gt * self.d1
        self.Zq = None
        if Zq is not None:
            self.Zq, self.yq = np.asarray(Zq, np.float64), np.asarray(yq, np.float64)
            self.Zqt = (self.Zq - self.mu) / self.ell

    # ---- H u ------------------------------------------------------------------------------------------------------
    def hvp(self, u, want_feature_part=False):
        n, s = self.n, self.s
        ut = np.asarray(u, np.float64) * self.d1
        un, us, ul = ut[0], ut[1], ut[2:]
        c = -ul / self.ell
        Ddot = 2.0 * _weighted_sqdist(self.Zt, c)
        Ainv, alpha = self.inner["Ainv"], self.inner["alpha"]
        Adot = un * np.eye(n) + us * self.k0 + s * self.k1 * Ddot
        X = Ainv @ Adot
        Y = X @ Ainv
        adot = -X @ alpha
        Qdot = 0.5 * (-Y - np.outer(adot, alpha) - np.outer(alpha, adot))
        Ht_n = np.trace(Qdot) / n - self.d2pn * un / n
        Ht_s = ((Qdot * self.k0).sum() + (self.Q * self.k1 * Ddot).sum()) / n
        Wdot = (Qdot * s * self.k1 + self.Q * (us * self.k1 + s * self.k2 * Ddot)) / n
        Gdot = dz_from_weights(self.Zt, None, Wdot, None, None)[0] + dz_from_weights(self.Zt * c, None, self.W, None, None)[0]
        S2 = (self.Zt * Gdot).sum(0)
        Ht_l = 2.0 * ul * self.S1 / self.ell ** 2 - S2 / self.ell - self.d2pl * ul / n
        Ht = np.concatenate([[Ht_n, Ht_s], Ht_l])
        Hu = self.d1 * Ht + self.gt * self.d2 * np.asarray(u, np.float64)
        if want_feature_part:
            return Hu, Gdot + c * self.G      # d (u^T grad_phi f_in) / d Z~
        return Hu

    def hessian(self):
        h = 2 + self.d
        return np.stack([self.hvp(np.eye(h)[a]) for a in range(h)], 1)

    # ---- outer ------------------------------------------------------------------------------------------------------
    def outer(self):
        D2qs, D2qq = sqdist(self.Zqt, self.Zt), sqdist(self.Zqt, self.Zqt)
        inner = dict(self.inner)
        o = outer_stage(D2qs, D2qq, self.yq, self.ys, inner, self.kind)
        dZs, dZq = dz_from_weights(self.Zt, self.Zqt, o["W_ss"], o["W_qs"], o["W_qq"])
        g_l = -((self.Zt * dZs).sum(0) + (self.Zqt * dZq).sum(0)) / self.ell
        g_out = np.concatenate([o["g_out"][:2], g_l * self.d1[2:]])
        return dict(f_out=o["f_out"], g_out=g_out, dZst=dZs, dZqt=dZq, mean=o["mean"], cov=o["cov"])


def cg_solve(hvp, b, tol=1e-12, maxiter=None):
    """Plain conjugate gradients (what the HIP path runs, there in float32 with a looser tolerance)."""
    x = np.zeros_like(b)
    r = b.copy()
    p = r.copy()
    rs = r @ r
    b2 = b @ b
    for it in range(maxiter or 10 * len(b)):
        if rs <= tol * tol * b2:
            break
        Hp = hvp(p)
        a = rs / (p @ Hp)
        x += a * p
        r -= a * Hp
        rs_new = r @ r
        p = r + (rs_new / rs) * p
        rs = rs_new
    return x, it


def full_pipeline_ard(Zs, ys, Zq, yq, phi, pri, kind, dense_solve=True):
    t = ArdTask(Zs, ys, phi, pri, kind, Zq, yq)
    o = t.outer()
    if dense_solve:
        H = t.hessian()
        v = np.linalg.solve(H, o["g_out"])
    else:
        H = None
        v, _ = cg_solve(t.hvp, o["g_out"])
    _, mixed_t = t.hvp(v, want_feature_part=True)
    return dict(f_in=t.f_in, g_in=t.g_in, H=H, f_out=o["f_out"], g_out=o["g_out"], v=v,
                dfin_dZs=t.G / t.ell, dZs_direct=o["dZst"] / t.ell, dZq_direct=o["dZqt"] / t.ell, mixed_Zs=mixed_t / t.ell,
                dZs_total=(o["dZst"] - mixed_t) / t.ell, dZq_total=o["dZqt"] / t.ell, pred_mean=o["mean"],
                pred_var=np.diag(o["cov"]).copy())
