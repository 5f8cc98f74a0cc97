// ARD kernel (one lengthscale per feature dimension: fs_mol/models/adaptive_dkt.py:107-108, gp_utils.py:27-30),
// h = 2 + d inner parameters (raw_noise, raw_outputscale, raw_lengthscale[d]).
//
// The kernel depends on z_k / l_k only, so the whole non-ARD pipeline is reused on SCALED features
// z~ = (z - mean_s) / l at unit lengthscale, and (oracle/closed_form_ard.py)
//     d f / d l_k  = -(1 / l_k) sum_i z~_ik  d f / d z~_ik          (Euler homogeneity)
//     d f / d z_ik =  (1 / l_k) d f / d z~_ik
// A Hessian-vector product H u (u in raw-parameter space) is the directional derivative of (dF/dnoise, dF/ds, dF/dZ~)
// along (u_n, u_s, Zdot = Z~ * c), c_k = -u_lk / l_k:
//     Ddot = 2 sum_k c_k (z~_ik - z~_jk)^2                           [one N x N x d product]
//     Adot = u_n I + u_s kappa + s kappa' . Ddot,   X = Ainv Adot,  Y = X Ainv,  adot = -X alpha   [two N^3 products]
//     Qdot = (-Y - adot alpha^T - alpha adot^T) / 2,   Wdot = (Qdot . s kappa' + Q . (u_s kappa' + s kappa'' . Ddot)) / n
//     Gdot' = 4 (rowsum(Wdot) . Z~ - Wdot Z~)                        [one N x N x d product]
//     (H_t u)_lk = 3 u_lk S1_k / l_k^2 - S2'_k / l_k - prior''_k u_lk / n,   S1 = colsum(Z~ . G), S2' = colsum(Z~ . Gdot')
// so the IFT system H v = grad_phi f_out is solved by conjugate gradients without ever forming the h x h Hessian, and the
// mixed term d(v^T grad_phi f_in)/dZ~ = Gdot'(v) + 2 c(v) . G falls out of one more such pass.
// The inner fit is an L-BFGS (m = 10) with the same Armijo/interpolation line search as the 3-parameter BFGS of
// inner.h, one workgroup per task, state in the workspace; every evaluation re-scales the features and re-runs the
// non-ARD evaluation kernels.
#pragma once
#include "large.h"

namespace adkf {

constexpr int ARD_M = 10;            // L-BFGS history (SciPy maxcor default)
constexpr float RAW_ONE = 0.54132485461291810f;   // softplus(RAW_ONE) = 1

struct ArdFitState {
    int phase, evals, bt, hist, head;
    float f, step, gp, gamma;
    float rho[ARD_M];
};

struct ArdCgState { float rs, b2, pHp; int done, iters, breakdown; };

struct ArdView {
    int T, d, h, ns_ld, nq_ld;
    const int32_t *n_s, *n_q;
    const float *Z_s, *Z_q;      // original features
    float *Zt_s, *Zt_q;          // scaled features
    float *mu;                   // [T, d] support column means
    float *ell;                  // [T, d]
    float *phi3, *pri3;          // [T, 3], [T, 4] for the unit-lengthscale pipeline
    const float* priors;         // [T, 4] (the caller's)
    float *f3, *g3;              // [T], [T, 3] outputs of the unit-lengthscale evaluation
    float *S1;                   // [T, d] colsum(Z~ . G)
    float *gt;                   // [T, h] d f_in / d transformed
    __device__ __forceinline__ int ns(int t) const { return n_s ? n_s[t] : ns_ld; }
    __device__ __forceinline__ int nq(int t) const { return n_q ? n_q[t] : nq_ld; }
};

// block-wide reductions for 256 threads; contain barriers, every thread gets the result
__device__ __forceinline__ float bsum256(float v, float* red) {
    float a[1] = {v};
    block_sum<1, 256>(a, red);
    return a[0];
}
__device__ __forceinline__ float bmax256(float v, float* red) {
    v = wave_max(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

// ---- scaling ----------------------------------------------------------------------------------------------------------
// ell = softplus(x[2:]), phi3 = (x0, x1, RAW_ONE), pri3 = noise prior only;  grid: ceil(d / 256) x T
__global__ __launch_bounds__(256) void k_ard_params(ArdView a, const float* x) {
    const int t = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k < a.d) a.ell[(size_t)t * a.d + k] = softplus_f(x[(size_t)t * a.h + 2 + k]);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        a.phi3[t * 3 + 0] = x[(size_t)t * a.h]; a.phi3[t * 3 + 1] = x[(size_t)t * a.h + 1]; a.phi3[t * 3 + 2] = RAW_ONE;
        a.pri3[t * 4 + 0] = a.priors[t * 4 + 0]; a.pri3[t * 4 + 1] = a.priors[t * 4 + 1]; a.pri3[t * 4 + 2] = 0.f; a.pri3[t * 4 + 3] = -1.f;
    }
}

// Zt = (Z - mu) / ell (rows beyond n are zeroed): one wave per row, grid (ceil(ld / 4), T).  (The squared row norms the
// distance stage needs are summed by the distance GEMM itself while it stages Zt: gemm.h set_rowsq.)
__global__ __launch_bounds__(256) void k_ard_scale(ArdView a, const float* Z, float* Zt, const int32_t* n_arr, int ld) {
    const int t = blockIdx.y, i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= ld) return;
    const int n = n_arr ? n_arr[t] : ld;
    const size_t o = ((size_t)t * ld + i) * a.d;
    const float *mu = a.mu + (size_t)t * a.d, *el = a.ell + (size_t)t * a.d;
    for (int k = lane; k < a.d; k += 64) Zt[o + k] = i < n ? (Z[o + k] - mu[k]) / el[k] : 0.f;
}

// out[k] = sum_i A_ik B_ik over the rows of one task (64 columns x 4 row groups; grid: ceil(d / 64) x T); optional second
// pair (A2, B2) with its own row count is added (support + query)
struct ArdColdot { const float *A, *B; const int32_t* n_arr; int ld; const float *A2, *B2; const int32_t* n2_arr; int ld2; float* out; int d; };

__global__ __launch_bounds__(256) void k_ard_coldot(ArdColdot a) {
    __shared__ float part[4][64];
    const int t = blockIdx.y, cl = threadIdx.x & 63, g = threadIdx.x >> 6, k = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (k < a.d) {
        const int n = a.n_arr ? a.n_arr[t] : a.ld;
        const float *A = a.A + (size_t)t * a.ld * a.d, *B = a.B + (size_t)t * a.ld * a.d;
        for (int i = g; i < n; i += 4) s += A[(size_t)i * a.d + k] * B[(size_t)i * a.d + k];
        if (a.A2) {
            const int m = a.n2_arr ? a.n2_arr[t] : a.ld2;
            const float *A2 = a.A2 + (size_t)t * a.ld2 * a.d, *B2 = a.B2 + (size_t)t * a.ld2 * a.d;
            for (int i = g; i < m; i += 4) s += A2[(size_t)i * a.d + k] * B2[(size_t)i * a.d + k];
        }
    }
    part[g][cl] = s;
    __syncthreads();
    if (g == 0 && k < a.d) a.out[(size_t)t * a.d + k] = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}

// ---- value and gradient in the h raw parameters from the unit-lengthscale evaluation --------------------------------
// f = f3 - sum_k log p(l_k) / n;  gt_lk = -S1_k / l_k - p'(l_k) / n;  g = gt * sigmoid(raw).   One workgroup per task.
struct ArdEvalFin { ArdView v; const float* x; float* f; float* g; const int32_t* info3; };

__global__ __launch_bounds__(256) void k_ard_eval_fin(ArdEvalFin a) {
    __shared__ float red[4];
    const int t = blockIdx.x, tid = threadIdx.x, d = a.v.d, h = a.v.h;
    const float fn = (float)a.v.ns(t);
    const float* x = a.x + (size_t)t * h;
    const float *pr = a.v.priors + t * 4, *ell = a.v.ell + (size_t)t * d, *S1 = a.v.S1 + (size_t)t * d;
    float* gt = a.v.gt + (size_t)t * h;
    float* g = a.g + (size_t)t * h;
    float lp = 0.f;
    for (int k = tid; k < d; k += 256) {
        const float l = ell[k];
        float dp = 0.f;
        if (pr[3] > 0.f) {
            const float lx = logf(l), sc = pr[3], z = (lx - pr[2]) / (sc * sc);
            lp += -lx - logf(sc) - 0.5f * LOG_2PI - 0.5f * (lx - pr[2]) * z;
            dp = (-1.f - z) / l;
        }
        const float v = -S1[k] / l - dp / fn;
        gt[2 + k] = v;
        g[2 + k] = v * sigmoid_f(x[2 + k]);
    }
    lp = bsum256(lp, red);
    if (tid == 0) {
        const float g0 = a.v.g3[t * 3 + 0], g1 = a.v.g3[t * 3 + 1];
        g[0] = g0; g[1] = g1;
        gt[0] = g0 / sigmoid_f(x[0]); gt[1] = g1 / sigmoid_f(x[1]);
        float f = a.v.f3[t] - lp / fn;
        if (a.info3[t] != 0 || !(f == f)) f = INFINITY;
        a.f[t] = f;
    }
}

// ---- L-BFGS transition (one workgroup per task) -----------------------------------------------------------------------
struct ArdFitArgs {
    int T, h, max_evals, exact_evals;
    float gtol, ftol;
    ArdFitState* st;
    float *x, *g, *p, *xe, *ge;     // [T, h] each; ge / fe are the evaluation just made at xe
    float *S, *Y;                   // [T, ARD_M, h]
    const float* fe;
    const int32_t* info_eval;
    // outputs (written once, when the task finishes)
    float *phi, *f_final, *gnorm; int32_t *nevals, *info;
};

__global__ void k_ard_fit_begin(ArdFitArgs a) {
    const int t = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k < a.h) { const float v = a.phi[(size_t)t * a.h + k]; a.x[(size_t)t * a.h + k] = v; a.xe[(size_t)t * a.h + k] = v; }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        ArdFitState& s = a.st[t];
        s.phase = a.max_evals > 0 ? PH_INIT : PH_FINAL; s.evals = 0; s.bt = 0; s.hist = 0; s.head = 0;
        s.f = INFINITY; s.step = 1.f; s.gp = 0.f; s.gamma = 1.f;
    }
}

__global__ __launch_bounds__(256) void k_ard_advance(ArdFitArgs a) {
    __shared__ float red[4];
    __shared__ float alpha_s[ARD_M];
    const int t = blockIdx.x, tid = threadIdx.x, h = a.h;
    ArdFitState& S = a.st[t];
    int phase = S.phase;
    if (phase == PH_DONE) return;
    float *x = a.x + (size_t)t * h, *g = a.g + (size_t)t * h, *p = a.p + (size_t)t * h, *xe = a.xe + (size_t)t * h;
    const float* ge = a.ge + (size_t)t * h;
    float *Sv = a.S + (size_t)t * ARD_M * h, *Yv = a.Y + (size_t)t * ARD_M * h;
    const float fe = a.fe[t];
    const int ie = a.info_eval[t];
    const bool exact = a.exact_evals != 0;
    float gmax = 0.f;
    for (int k = tid; k < h; k += 256) gmax = fmaxf(gmax, fabsf(ge[k]));
    gmax = bmax256(gmax, red);
    if (phase == PH_FINAL) {
        for (int k = tid; k < h; k += 256) if (a.max_evals > 0) a.phi[(size_t)t * h + k] = xe[k];
        if (tid == 0) {
            a.info[t] = (ie != 0 || !(fe < INFINITY)) ? (ie != 0 ? ie : 1) : 0;
            if (a.f_final) a.f_final[t] = fe;
            if (a.gnorm) a.gnorm[t] = gmax;
            if (a.nevals) a.nevals[t] = S.evals + 1;
            S.phase = PH_DONE;
        }
        return;
    }
    const int evals = S.evals + 1, budget = a.max_evals - 1;
    float f = S.f, step = S.step, gp = S.gp, gamma = S.gamma;
    int bt = S.bt, hist = S.hist, head = S.head;
    bool stop = false, new_dir = false;
    if (phase == PH_INIT) {
        for (int k = tid; k < h; k += 256) { x[k] = xe[k]; g[k] = ge[k]; }
        f = fe;
        hist = 0; head = 0;
        if (ie != 0 || !(fe < INFINITY)) stop = true;
        else if (gmax <= a.gtol) stop = true;
        else new_dir = true;
        phase = PH_SEARCH;
    } else if (phase == PH_SEARCH) {
        if (ie == 0 && fe <= f + 1e-4f * step * gp) {  // Armijo (a failed factorisation never counts as progress)
            float sy = 0.f, yy = 0.f, ss = 0.f;
            for (int k = tid; k < h; k += 256) { const float s_ = xe[k] - x[k], y_ = ge[k] - g[k]; sy += s_ * y_; yy += y_ * y_; ss += s_ * s_; }
            sy = bsum256(sy, red); yy = bsum256(yy, red); ss = bsum256(ss, red);
            if (sy > 1e-10f * sqrtf(ss * yy) && yy > 0.f) {
                for (int k = tid; k < h; k += 256) { Sv[(size_t)head * h + k] = xe[k] - x[k]; Yv[(size_t)head * h + k] = ge[k] - g[k]; }
                if (tid == 0) S.rho[head] = 1.f / sy;
                head = (head + 1) % ARD_M;
                hist = hist < ARD_M ? hist + 1 : ARD_M;
                gamma = sy / yy;
            }
            const float fprev = f;
            for (int k = tid; k < h; k += 256) { x[k] = xe[k]; g[k] = ge[k]; }
            f = fe;
            if (gmax <= a.gtol || fabsf(fprev - fe) <= a.ftol * fmaxf(fmaxf(fabsf(fprev), fabsf(fe)), 1.f)) stop = true;
            else new_dir = true;
        } else {
            const float denom = 2.f * (fe - f - gp * step);
            const float sq = (denom > 0.f && fe < INFINITY) ? (-gp * step * step / denom) : 0.5f * step;
            step = fminf(fmaxf(sq, 0.1f * step), 0.5f * step);
            if (++bt >= 12) stop = true;  // converged to working precision (exact mode: burns the rest of the budget here)
        }
    }
    __syncthreads();  // S.rho[head] visible; x, g final
    if (new_dir && !stop) {
        // two-loop recursion: p = -H_k g
        for (int k = tid; k < h; k += 256) p[k] = g[k];
        for (int j = 0; j < hist; ++j) {
            const int idx = (head - 1 - j + 2 * ARD_M) % ARD_M;
            float s = 0.f;
            for (int k = tid; k < h; k += 256) s += Sv[(size_t)idx * h + k] * p[k];
            const float aj = S.rho[idx] * bsum256(s, red);
            for (int k = tid; k < h; k += 256) p[k] -= aj * Yv[(size_t)idx * h + k];
            if (tid == 0) alpha_s[j] = aj;
        }
        __syncthreads();
        const float sc = hist > 0 ? gamma : 1.f;
        for (int k = tid; k < h; k += 256) p[k] *= sc;
        for (int j = hist - 1; j >= 0; --j) {
            const int idx = (head - 1 - j + 2 * ARD_M) % ARD_M;
            float s = 0.f;
            for (int k = tid; k < h; k += 256) s += Yv[(size_t)idx * h + k] * p[k];
            const float b = S.rho[idx] * bsum256(s, red);
            const float cf = alpha_s[j] - b;
            for (int k = tid; k < h; k += 256) p[k] += cf * Sv[(size_t)idx * h + k];
        }
        float dg = 0.f, g1 = 0.f, gg = 0.f, pm = 0.f, gm = 0.f;
        for (int k = tid; k < h; k += 256) {
            p[k] = -p[k]; dg += g[k] * p[k]; g1 += fabsf(g[k]); gg += g[k] * g[k];
            pm = fmaxf(pm, fabsf(p[k])); gm = fmaxf(gm, fabsf(g[k]));
        }
        dg = bsum256(dg, red); g1 = bsum256(g1, red); gg = bsum256(gg, red); pm = bmax256(pm, red); gm = bmax256(gm, red);
        if (!(dg < 0.f)) {  // not a descent direction (or NaN): restart from steepest descent
            hist = 0;
            for (int k = tid; k < h; k += 256) p[k] = -g[k];
            dg = -gg;
            pm = gm;
            if (!(dg < 0.f)) stop = true;
        }
        gp = dg;
        step = hist > 0 ? 1.f : fminf(1.f, 1.f / g1);
        step = fminf(step, MAX_MOVE / fmaxf(pm, 1e-30f));   // as Bfgs::direction: no raw parameter moves more than MAX_MOVE per trial
        bt = 0;
    }
    if (stop && exact) phase = PH_BURN;
    if (evals >= budget || (stop && !exact)) {
        phase = PH_FINAL;
        for (int k = tid; k < h; k += 256) xe[k] = x[k];
    } else if (phase == PH_BURN) {
        for (int k = tid; k < h; k += 256) xe[k] = x[k];
    } else {
        for (int k = tid; k < h; k += 256) xe[k] = x[k] + step * p[k];
    }
    if (tid == 0) {
        S.phase = phase; S.evals = evals; S.bt = bt; S.hist = hist; S.head = head;
        S.f = f; S.step = step; S.gp = gp; S.gamma = gamma;
    }
}

// ---- Hessian-vector product ---------------------------------------------------------------------------------------------
struct ArdHvp {
    ArdView v;
    TaskView tv;                 // of the scaled batch
    const float* x;              // [T, h] raw parameters the Hessian is taken at
    const float* u;              // [T, h] direction
    float* Hu;                   // [T, h]
    float *c;                    // [T, d]
    float *ut2;                  // [T, 2] transformed (u_n, u_s)
    float *wn;                   // [T, ns] weighted row norms
    const float *D2, *Ainv;      // [T, ns, ns]
    float *Ddot, *X, *Wdot;      // [T, ns, ns]
    float *adot;                 // [T, ns]
    float *part; int ntiles;     // [T, ntiles, 4]
    const float* G;              // [T, ns, d]  d f_in / d Z~
    float* Gdot;                 // [T, ns, d]
    float* S2;                   // [T, d]
    const ArdCgState* cg;        // optional: tasks with cg[t].done skip (their Hu is not used)
};

// c_k = -u_k sigmoid(x_k) / l_k;  ut2 = (u_0 sigmoid(x_0), u_1 sigmoid(x_1));  grid: ceil(d / 256) x T
__global__ __launch_bounds__(256) void k_ard_dir(ArdHvp a) {
    const int t = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x, h = a.v.h;
    const float *x = a.x + (size_t)t * h, *u = a.u + (size_t)t * h;
    if (k < a.v.d) a.c[(size_t)t * a.v.d + k] = -u[2 + k] * sigmoid_f(x[2 + k]) / a.v.ell[(size_t)t * a.v.d + k];
    if (blockIdx.x == 0 && threadIdx.x == 0) { a.ut2[t * 2] = u[0] * sigmoid_f(x[0]); a.ut2[t * 2 + 1] = u[1] * sigmoid_f(x[1]); }
}

// wn_i = sum_k 2 c_k z~_ik^2  (wave per row; grid: ceil(ns_ld / 4) x T)
__global__ __launch_bounds__(256) void k_ard_wnorm(ArdHvp a) {
    const int t = blockIdx.y, i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, d = a.v.d;
    if (i >= a.v.ns_ld) return;
    float s = 0.f;
    if (i < a.v.ns(t)) {
        const float* z = a.v.Zt_s + ((size_t)t * a.v.ns_ld + i) * d;
        const float* c = a.c + (size_t)t * d;
        for (int k = lane; k < d; k += 64) s += 2.f * c[k] * z[k] * z[k];
    }
    s = wave_sum(s);
    if (lane == 0) a.wn[(size_t)t * a.v.ns_ld + i] = s;
}

// Ddot_ij = wn_i + wn_j - 2 sum_k (2 c_k z~_ik) z~_jk
struct ProbArdDdot {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = true;
    static constexpr int NRED = 0;
    ArdHvp h;
    int n, d; const float *Z, *c, *wn; float* Do; bool vec;
    __device__ bool setup(int t) {
        n = h.v.ns(t); d = h.v.d; vec = h.tv.vec;
        Z = h.v.Zt_s + (size_t)t * h.v.ns_ld * d; c = h.c + (size_t)t * d; wn = h.wn + (size_t)t * h.v.ns_ld;
        Do = h.Ddot + (size_t)t * h.v.ns_ld * h.v.ns_ld;
        return n > 0;
    }
    __device__ int M() const { return n; } __device__ int N() const { return n; } __device__ int K() const { return d; }
    __device__ float a(int i, int k) const { return 2.f * c[k] * Z[(size_t)i * d + k]; }
    __device__ float b(int k, int j) const { return Z[(size_t)j * d + k]; }
    __device__ void a4(int i, int k, float (&v)[4]) const {
        float c4[4];
        ld4(Z + (size_t)i * d + k, v); ld4(c + k, c4);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] *= 2.f * c4[q];
    }
    __device__ void b4(int k, int j, float (&v)[4]) const { ld4(Z + (size_t)j * d + k, v); }
    __device__ void epi(int i, int j, float acc, float*) const { Do[(size_t)i * h.v.ns_ld + j] = (i == j) ? 0.f : wn[i] + wn[j] - 2.f * acc; }
    __device__ void store_red(int, const float*) const {}
};

// X = Ainv * Adot,  Adot = u_n I + u_s kappa(D2) + s kappa'(D2) . Ddot  (symmetric: row j read contiguously in k)
struct ProbArdX {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = true;
    static constexpr int NRED = 0;
    ArdHvp h;
    int n, ld; float un, us, os; const float *Ai, *D2, *Dd; float* Xo; bool vec;
    __device__ bool setup(int t) {
        n = h.v.ns(t); ld = h.v.ns_ld; vec = h.tv.vec;
        un = h.ut2[t * 2]; us = h.ut2[t * 2 + 1]; os = h.tv.scal[(size_t)t * NSCAL + S_OS];
        Ai = h.Ainv + (size_t)t * ld * ld; D2 = h.D2 + (size_t)t * ld * ld; Dd = h.Ddot + (size_t)t * ld * ld; Xo = h.X + (size_t)t * ld * ld;
        return n > 0;
    }
    __device__ int M() const { return n; } __device__ int N() const { return n; } __device__ int K() const { return n; }
    __device__ float adot(int j, int k, float d2, float dd) const {
        float k0, k1, k2; kappa3(h.tv.kind, d2, k0, k1, k2);
        return (j == k ? un : 0.f) + us * k0 + os * k1 * dd;
    }
    __device__ float a(int i, int k) const { return Ai[(size_t)i * ld + k]; }
    __device__ float b(int k, int j) const { return adot(j, k, D2[(size_t)j * ld + k], Dd[(size_t)j * ld + k]); }
    __device__ void a4(int i, int k, float (&v)[4]) const { ld4(Ai + (size_t)i * ld + k, v); }
    __device__ void b4(int k, int j, float (&v)[4]) const {
        float dd[4];
        ld4(D2 + (size_t)j * ld + k, v); ld4(Dd + (size_t)j * ld + k, dd);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = adot(j, k + q, v[q], dd[q]);
    }
    __device__ void epi(int i, int j, float acc, float*) const { Xo[(size_t)i * ld + j] = acc; }
    __device__ void store_red(int, const float*) const {}
};

// adot = -X alpha  (wave per row; grid: ceil(ns_ld / 4) x T)
__global__ __launch_bounds__(256) void k_ard_adot(ArdHvp a) {
    const int t = blockIdx.y, i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, ld = a.v.ns_ld;
    const int n = a.v.ns(t);
    if (i >= n) return;
    const float* row = a.X + ((size_t)t * ld + i) * ld;
    const float* al = a.tv.vec_ptr(t, V_ALPHA);
    float s = 0.f;
    for (int j = lane; j < n; j += 64) s += row[j] * al[j];
    s = wave_sum(s);
    if (lane == 0) a.adot[(size_t)t * ld + i] = -s;
}

// Y = X Ainv; epilogue: Qdot, Wdot and the three reductions tr(Qdot), <Qdot, kappa>, <Q, kappa' . Ddot>
struct ProbArdY {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = true;
    static constexpr int NRED = 3;
    ArdHvp h;
    int n, ld, t_; float us, os, fn; const float *Ai, *D2, *Dd, *Xi, *al, *ad; float* Wo; bool vec;
    __device__ bool setup(int t) {
        t_ = t; n = h.v.ns(t); ld = h.v.ns_ld; vec = h.tv.vec; fn = (float)n;
        us = h.ut2[t * 2 + 1]; os = h.tv.scal[(size_t)t * NSCAL + S_OS];
        Ai = h.Ainv + (size_t)t * ld * ld; D2 = h.D2 + (size_t)t * ld * ld; Dd = h.Ddot + (size_t)t * ld * ld; Xi = h.X + (size_t)t * ld * ld;
        Wo = h.Wdot + (size_t)t * ld * ld; al = h.tv.vec_ptr(t, V_ALPHA); ad = h.adot + (size_t)t * ld;
        return n > 0;
    }
    __device__ int M() const { return n; } __device__ int N() const { return n; } __device__ int K() const { return n; }
    __device__ float a(int i, int k) const { return Xi[(size_t)i * ld + k]; }
    __device__ float b(int k, int j) const { return Ai[(size_t)j * ld + k]; }
    __device__ void a4(int i, int k, float (&v)[4]) const { ld4(Xi + (size_t)i * ld + k, v); }
    __device__ void b4(int k, int j, float (&v)[4]) const { ld4(Ai + (size_t)j * ld + k, v); }
    __device__ void epi(int i, int j, float acc, float* red) const {
        const float qd = 0.5f * (-acc - ad[i] * al[j] - al[i] * ad[j]);
        const float q = 0.5f * (Ai[(size_t)i * ld + j] - al[i] * al[j]);
        float k0, k1, k2; kappa3(h.tv.kind, D2[(size_t)i * ld + j], k0, k1, k2);
        const float dd = Dd[(size_t)i * ld + j];
        Wo[(size_t)i * ld + j] = (qd * os * k1 + q * (us * k1 + os * k2 * dd)) / fn;
        if (i == j) red[0] += qd;
        red[1] += qd * k0;
        red[2] += q * k1 * dd;
    }
    __device__ void store_red(int tile, const float* red) const {
        float* p = h.part + ((size_t)t_ * h.ntiles + tile) * 4;
        p[0] = red[0]; p[1] = red[1]; p[2] = red[2];
    }
};

// Hu from the pieces (one workgroup per task):
//   Ht_n = tr(Qdot)/n - prior''(noise) u_n / n;  Ht_s = (<Qdot, kappa> + <Q, kappa' Ddot>) / n
//   Ht_lk = 3 u_lk S1_k / l_k^2 - S2'_k / l_k - prior''(l_k) u_lk / n;      Hu = d1 Ht + gt d2 u
struct ArdHvpFin { ArdHvp h; };

__global__ __launch_bounds__(256) void k_ard_hvp_fin(ArdHvp a) {
    const int t = blockIdx.x, tid = threadIdx.x, d = a.v.d, h = a.v.h;
    if (a.cg && a.cg[t].done) return;
    const float fn = (float)a.v.ns(t);
    const float *x = a.x + (size_t)t * h, *u = a.u + (size_t)t * h, *gt = a.v.gt + (size_t)t * h;
    const float *pr = a.v.priors + t * 4, *ell = a.v.ell + (size_t)t * d, *S1 = a.v.S1 + (size_t)t * d, *S2 = a.S2 + (size_t)t * d;
    float* Hu = a.Hu + (size_t)t * h;
    for (int k = tid; k < d; k += 256) {
        const float l = ell[k], sg = sigmoid_f(x[2 + k]), ul = u[2 + k] * sg;
        float d2p = 0.f;
        if (pr[3] > 0.f) { const float lx = logf(l), s2 = pr[3] * pr[3]; d2p = (1.f + (lx - pr[2]) / s2 - 1.f / s2) / (l * l); }
        const float ht = 3.f * ul * S1[k] / (l * l) - S2[k] / l - d2p * ul / fn;
        Hu[2 + k] = sg * ht + gt[2 + k] * sg * (1.f - sg) * u[2 + k];
    }
    if (tid == 0) {
        float p0 = 0.f, p1 = 0.f, p2 = 0.f;
        for (int q = 0; q < a.ntiles; ++q) { const float* p = a.part + ((size_t)t * a.ntiles + q) * 4; p0 += p[0]; p1 += p[1]; p2 += p[2]; }
        const float* sc = a.tv.scal + (size_t)t * NSCAL;
        const float noise = sc[S_NOISE];
        const float sn = sigmoid_f(x[0]), ss = sigmoid_f(x[1]);
        const float un = u[0] * sn;
        float d2pn = 0.f;
        if (pr[1] > 0.f) { const float lx = logf(noise), s2 = pr[1] * pr[1]; d2pn = (1.f + (lx - pr[0]) / s2 - 1.f / s2) / (noise * noise); }
        const float htn = p0 / fn - d2pn * un / fn;
        const float hts = (p1 + p2) / fn;
        Hu[0] = sn * htn + gt[0] * sn * (1.f - sn) * u[0];
        Hu[1] = ss * hts + gt[1] * ss * (1.f - ss) * u[1];
    }
}

// ---- conjugate gradients (one workgroup per task) -------------------------------------------------------------------------
struct ArdCg { int T, h; float tol; ArdCgState* st; const float* b; float *x, *r, *p; const float* Hp;
               const int32_t* n_s; int ns_ld; int32_t* n_eff; };  // n_eff[t] = 0 once task t has converged: every kernel of the next HVP skips it

__global__ __launch_bounds__(256) void k_ard_cg_begin(ArdCg a) {
    __shared__ float red[4];
    const int t = blockIdx.x, tid = threadIdx.x, h = a.h;
    float s = 0.f;
    for (int k = tid; k < h; k += 256) {
        const float v = a.b[(size_t)t * h + k];
        a.x[(size_t)t * h + k] = 0.f; a.r[(size_t)t * h + k] = v; a.p[(size_t)t * h + k] = v;
        s += v * v;
    }
    s = bsum256(s, red);
    if (tid == 0) {
        ArdCgState& c = a.st[t]; c.rs = s; c.b2 = s; c.pHp = 0.f; c.done = (s == 0.f) ? 1 : 0; c.iters = 0; c.breakdown = 0;
        a.n_eff[t] = c.done ? 0 : (a.n_s ? a.n_s[t] : a.ns_ld);
    }
}

__global__ __launch_bounds__(256) void k_ard_cg_step(ArdCg a) {
    __shared__ float red[4];
    const int t = blockIdx.x, tid = threadIdx.x, h = a.h;
    ArdCgState& c = a.st[t];
    if (c.done) return;
    float *x = a.x + (size_t)t * h, *r = a.r + (size_t)t * h, *p = a.p + (size_t)t * h;
    const float* Hp = a.Hp + (size_t)t * h;
    float pHp = 0.f;
    for (int k = tid; k < h; k += 256) pHp += p[k] * Hp[k];
    pHp = bsum256(pHp, red);
    const float rs = c.rs;
    if (!(pHp > 0.f)) {  // negative curvature or NaN: H is not positive definite along p - keep the current iterate
        if (tid == 0) { c.done = 1; c.breakdown = 1; a.n_eff[t] = 0; }
        return;
    }
    const float al = rs / pHp;
    float rn = 0.f;
    for (int k = tid; k < h; k += 256) { x[k] += al * p[k]; const float v = r[k] - al * Hp[k]; r[k] = v; rn += v * v; }
    rn = bsum256(rn, red);
    const float beta = rn / rs;
    for (int k = tid; k < h; k += 256) p[k] = r[k] + beta * p[k];
    if (tid == 0) { c.rs = rn; c.pHp = pHp; c.iters += 1; if (rn <= a.tol * a.tol * c.b2) { c.done = 1; a.n_eff[t] = 0; } }
}

// ---- outer gradient in the h raw parameters, final feature gradients ------------------------------------------------------
// g_out = (g3_n, g3_s, -(colsum(Z~s . dZ~s) + colsum(Z~q . dZ~q)) / l * sigmoid(raw));  grid: ceil(d / 256) x T
struct ArdGout { ArdView v; const float* x; const float* coldot; const float* g3; float* g_out; };

__global__ __launch_bounds__(256) void k_ard_gout(ArdGout a) {
    const int t = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x, h = a.v.h;
    if (k < a.v.d) a.g_out[(size_t)t * h + 2 + k] = -a.coldot[(size_t)t * a.v.d + k] / a.v.ell[(size_t)t * a.v.d + k] * sigmoid_f(a.x[(size_t)t * h + 2 + k]);
    if (blockIdx.x == 0 && threadIdx.x == 0) { a.g_out[(size_t)t * h] = a.g3[t * 3]; a.g_out[(size_t)t * h + 1] = a.g3[t * 3 + 1]; }
}

// dZ = (direct - corr * (Gdot' + 2 c . G)) / l   (support; corr = 0 or G == null: first-order only), dZ = direct / l (query)
struct ArdDzFin { ArdView v; const float* direct; const float* Gdot; const float* G; const float* c; float corr; float* dZ; const int32_t* n_arr; int ld; };

__global__ __launch_bounds__(256) void k_ard_dz_fin(ArdDzFin a) {
    const int t = blockIdx.z, i = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k >= a.v.d) return;
    const int n = a.n_arr ? a.n_arr[t] : a.ld;
    const size_t o = ((size_t)t * a.ld + i) * a.v.d + k;
    float v = 0.f;
    if (i < n) {
        v = a.direct[o];
        if (a.Gdot && a.corr != 0.f) v -= a.corr * (a.Gdot[o] + 2.f * a.c[(size_t)t * a.v.d + k] * a.G[o]);
        v /= a.v.ell[(size_t)t * a.v.d + k];
    }
    a.dZ[o] = v;
}

}  // namespace adkf
