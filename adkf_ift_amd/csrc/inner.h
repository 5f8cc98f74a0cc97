// k_inner: one workgroup per task.  Evaluates f_inner = -MLL/N with its exact gradient at phi
// (oracle/closed_form.py::inner_stage) and, in fit mode, runs the whole quasi-Newton inner optimisation
// (the reference's host-side SciPy L-BFGS-B, fs_mol/utils/adaptive_dkt_utils.py:91) inside this one launch:
// the squared-distance matrix stays in registers for the entire fit, the kernel matrix / factor / inverse
// live in LDS, and nothing touches the host between iterations.
#pragma once
#include "factor.h"

namespace adkf {

struct InnerArgs {
    const float* D2ss;   // [T, ld, ld]
    const float* y_s;    // [T, ld]
    const int32_t* n_s;  // [T] or null
    float* phi;          // [T, 3] in (and out in fit mode)
    const float* priors; // [T, 4]
    float* Ainv;         // [T, ld, ld] or null
    float* vecs;         // [T, NVEC, vld] (alpha) or null
    float* scal;         // [T, NSCAL] or null
    float* f_out;        // [T] or null
    float* g_out;        // [T, 3] or null
    float* gnorm_out;    // [T] or null
    int32_t* nevals_out; // [T] or null
    int32_t* info;       // [T]
    int T, ld, vld, kind;
    int max_evals;       // 0: single evaluation at phi; > 0: fit
    int exact_evals;
    float gtol, ftol;
};

template <int NMAX, int NT>
struct InnerSmem {
    float buf0[FactorShape<NMAX>::ELEMS];
    float buf1[FactorShape<NMAX>::ELEMS];
    float y[NMAX];
    float w[NMAX];
    float alpha[NMAX];
    float dinv[NMAX];
    float red[8 * (NT / 64)];
};

// One evaluation.  d2r = this thread's register-resident slice of D2 (element e = r * NT + tid).
// Returns f and raw-space gradient g[3]; leaves A^-1 in sm.buf0 (full symmetric), alpha in sm.alpha.
template <int NMAX, int NT, int KIND>
__device__ __forceinline__ int inner_eval(InnerSmem<NMAX, NT>& sm, const float (&d2r)[NMAX * NMAX / NT], int n,
                                          const float* x, const float* pri, float& f, float* g, float* extra) {
    constexpr int LD = FactorShape<NMAX>::LD;
    constexpr int EPT = NMAX * NMAX / NT;
    const int tid = threadIdx.x;
    const float noise = softplus_f(x[0]) + NOISE_LB, os = softplus_f(x[1]), ls = softplus_f(x[2]);
    const float d1n = sigmoid_f(x[0]), d1s = sigmoid_f(x[1]), d1l = sigmoid_f(x[2]);
    const float il2 = 1.f / (ls * ls);

    __syncthreads();  // previous users of buf0 are done
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
        const int e = r * NT + tid;
        const int i = e / NMAX, j = e % NMAX;
        if (i < n && j <= i) {
            float k0, k1, k2;
            kappa3<KIND>(d2r[r] * il2, k0, k1, k2);
            sm.buf0[i * LD + j] = os * k0 + (i == j ? noise : 0.f);
        }
    }
    float logdet;
    const int info = ldl_sweep<NMAX, NT>(sm.buf0, sm.buf1, sm.dinv, n, logdet, sm.red);
    // w = Y y ; alpha = Y^T w ; y^T alpha = |w|^2 ; tr(A^-1) = |Y|_F^2
    if (tid < n) {
        float s = 0.f;
        for (int j = 0; j <= tid; ++j) s += sm.buf1[tid * LD + j] * sm.y[j];
        sm.w[tid] = s;
    }
    __syncthreads();
    if (tid < n) {
        float s = 0.f;
        for (int k = tid; k < n; ++k) s += sm.buf1[k * LD + tid] * sm.w[k];
        sm.alpha[tid] = s;
    }
    ata_lower<NMAX, NT>(sm.buf1, sm.buf0, n);
    __syncthreads();
    // traces
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};  // tr(Ainv G), a^T G a, tr(Ainv), a^T a, y^T a
    const float gl = -2.f / ls;
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
        const int e = r * NT + tid;
        const int i = e / NMAX, j = e % NMAX;
        if (i < n && j < n) {
            float k0, k1, k2;
            const float u = d2r[r] * il2;
            kappa3<KIND>(u, k0, k1, k2);
            const float G = os * k1 * u * gl;
            acc[0] += sm.buf0[i * LD + j] * G;
            acc[1] += sm.alpha[i] * sm.alpha[j] * G;
        }
    }
    if (tid < n) {
        acc[2] = sm.buf0[tid * LD + tid];
        acc[3] = sm.alpha[tid] * sm.alpha[tid];
        acc[4] = sm.y[tid] * sm.alpha[tid];
    }
    block_sum<5, NT>(acc, sm.red);
    const float trAinvG = acc[0], aGa = acc[1], trAinv = acc[2], aa = acc[3], ya = acc[4];
    const float fn = (float)n;
    const float nll = 0.5f * ya + 0.5f * logdet + 0.5f * fn * LOG_2PI;
    // LogNormal priors on the transformed values (oracle/closed_form.py::lognormal_terms)
    float lp = 0.f, dpn = 0.f, dpl = 0.f;
    if (pri[1] > 0.f) {
        const float lx = logf(noise), sc = pri[1], z = (lx - pri[0]) / (sc * sc);
        lp += -lx - logf(sc) - 0.5f * LOG_2PI - 0.5f * (lx - pri[0]) * z;
        dpn = (-1.f - z) / noise;
    }
    if (pri[3] > 0.f) {
        const float lx = logf(ls), sc = pri[3], z = (lx - pri[2]) / (sc * sc);
        lp += -lx - logf(sc) - 0.5f * LOG_2PI - 0.5f * (lx - pri[2]) * z;
        dpl = (-1.f - z) / ls;
    }
    f = (nll - lp) / fn;
    const float gt0 = 0.5f * trAinv - 0.5f * aa - dpn;
    const float gt1 = (0.5f * (fn - noise * trAinv) - 0.5f * (ya - noise * aa)) / os;
    const float gt2 = 0.5f * trAinvG - 0.5f * aGa - dpl;
    g[0] = gt0 * d1n / fn;
    g[1] = gt1 * d1s / fn;
    g[2] = gt2 * d1l / fn;
    if (extra) {
        extra[0] = logdet; extra[1] = trAinv; extra[2] = aa; extra[3] = ya; extra[4] = trAinvG; extra[5] = aGa;
        extra[6] = gt0; extra[7] = gt1; extra[8] = gt2;
    }
    if (info != 0 || !(f == f)) {
        f = INFINITY;
        return info != 0 ? info : n + 1;
    }
    return 0;
}

template <int NMAX, int NT, int KIND>
__global__ __launch_bounds__(NT) void k_inner(InnerArgs a) {
    constexpr int LD = FactorShape<NMAX>::LD;
    constexpr int EPT = NMAX * NMAX / NT;
    __shared__ InnerSmem<NMAX, NT> sm;
    int t, tile;
    if (!task_tile(a.T, 1, t, tile)) return;
    const int tid = threadIdx.x;
    const int n = a.n_s ? a.n_s[t] : a.ld;
    const float* D2 = a.D2ss + (size_t)t * a.ld * a.ld;

    float d2r[EPT];
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
        const int e = r * NT + tid;
        const int i = e / NMAX, j = e % NMAX;
        d2r[r] = (i < n && j < n) ? D2[(size_t)i * a.ld + j] : 0.f;
    }
    if (tid < NMAX) sm.y[tid] = (tid < n) ? a.y_s[(size_t)t * a.ld + tid] : 0.f;
    float pri[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) pri[q] = a.priors[t * 4 + q];
    float x[3] = {a.phi[t * 3 + 0], a.phi[t * 3 + 1], a.phi[t * 3 + 2]};
    __syncthreads();

    float f, g[3], extra[9];
    int evals = 0;
    int info = 0;

    if (a.max_evals > 0) {
        // ---------------- quasi-Newton (BFGS, backtracking Armijo) on 3 raw parameters ----------------
        const int budget = a.max_evals - 1;  // the last evaluation is the output evaluation below
        info = inner_eval<NMAX, NT, KIND>(sm, d2r, n, x, pri, f, g, nullptr);
        ++evals;
        float Hi[3][3] = {{1.f, 0.f, 0.f}, {0.f, 1.f, 0.f}, {0.f, 0.f, 1.f}};
        bool first = true, done = (info != 0);
        while (!done && evals < budget) {
            float gmax = fmaxf(fabsf(g[0]), fmaxf(fabsf(g[1]), fabsf(g[2])));
            if (!a.exact_evals && gmax <= a.gtol) break;
            float p[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) p[i] = -(Hi[i][0] * g[0] + Hi[i][1] * g[1] + Hi[i][2] * g[2]);
            float gp = g[0] * p[0] + g[1] * p[1] + g[2] * p[2];
            if (!(gp < 0.f)) {  // not a descent direction (or NaN): restart from steepest descent
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) Hi[i][j] = (i == j) ? 1.f : 0.f;
#pragma unroll
                for (int i = 0; i < 3; ++i) p[i] = -g[i];
                gp = -(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
                first = true;
                if (!(gp < 0.f)) break;  // zero gradient
            }
            float step = 1.f;
            if (first) step = fminf(1.f, 1.f / (fabsf(g[0]) + fabsf(g[1]) + fabsf(g[2])));
            float fn_ = f, gn[3], xn[3];
            bool accepted = false;
            for (int bt = 0; bt < 12 && evals < budget; ++bt) {
#pragma unroll
                for (int i = 0; i < 3; ++i) xn[i] = x[i] + step * p[i];
                inner_eval<NMAX, NT, KIND>(sm, d2r, n, xn, pri, fn_, gn, nullptr);
                ++evals;
                if (fn_ <= f + 1e-4f * step * gp) { accepted = true; break; }
                // safeguarded quadratic interpolation
                float denom = 2.f * (fn_ - f - gp * step);
                float st = (denom > 0.f && fn_ < INFINITY) ? (-gp * step * step / denom) : 0.5f * step;
                step = fminf(fmaxf(st, 0.1f * step), 0.5f * step);
            }
            if (!accepted) {
                if (!a.exact_evals) break;  // line search failed: converged to working precision
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j) Hi[i][j] = (i == j) ? 1.f : 0.f;
                first = true;
                continue;
            }
            float s[3], yv[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) { s[i] = xn[i] - x[i]; yv[i] = gn[i] - g[i]; }
            const float sy = s[0] * yv[0] + s[1] * yv[1] + s[2] * yv[2];
            const float yy = yv[0] * yv[0] + yv[1] * yv[1] + yv[2] * yv[2];
            const float ss = s[0] * s[0] + s[1] * s[1] + s[2] * s[2];
            const float fprev = f;
            f = fn_;
#pragma unroll
            for (int i = 0; i < 3; ++i) { x[i] = xn[i]; g[i] = gn[i]; }
            if (sy > 1e-10f * sqrtf(ss * yy) && yy > 0.f) {
                if (first) {
                    const float sc = sy / yy;
#pragma unroll
                    for (int i = 0; i < 3; ++i)
#pragma unroll
                        for (int j = 0; j < 3; ++j) Hi[i][j] = (i == j) ? sc : 0.f;
                    first = false;
                }
                const float rho = 1.f / sy;
                float Hy[3];
#pragma unroll
                for (int i = 0; i < 3; ++i) Hy[i] = Hi[i][0] * yv[0] + Hi[i][1] * yv[1] + Hi[i][2] * yv[2];
                const float yHy = yv[0] * Hy[0] + yv[1] * Hy[1] + yv[2] * Hy[2];
#pragma unroll
                for (int i = 0; i < 3; ++i)
#pragma unroll
                    for (int j = 0; j < 3; ++j)
                        Hi[i][j] += -rho * (s[i] * Hy[j] + Hy[i] * s[j]) + rho * (rho * yHy + 1.f) * s[i] * s[j];
            }
            if (!a.exact_evals && fabsf(fprev - f) <= a.ftol * fmaxf(fmaxf(fabsf(fprev), fabsf(f)), 1.f)) break;
        }
        if (a.exact_evals) {  // burn the remaining budget at the current point: deterministic work
            while (evals < budget) {
                float fd, gd[3];
                inner_eval<NMAX, NT, KIND>(sm, d2r, n, x, pri, fd, gd, nullptr);
                ++evals;
            }
        }
    }
    // ---------------- output evaluation at the final point ----------------
    info = inner_eval<NMAX, NT, KIND>(sm, d2r, n, x, pri, f, g, extra);
    ++evals;

    if (a.max_evals > 0 && tid < 3) a.phi[t * 3 + tid] = x[tid];
    if (tid == 0) {
        a.info[t] = info;
        if (a.f_out) a.f_out[t] = f;
        if (a.g_out) { a.g_out[t * 3 + 0] = g[0]; a.g_out[t * 3 + 1] = g[1]; a.g_out[t * 3 + 2] = g[2]; }
        if (a.gnorm_out) a.gnorm_out[t] = fmaxf(fabsf(g[0]), fmaxf(fabsf(g[1]), fabsf(g[2])));
        if (a.nevals_out) a.nevals_out[t] = evals;
        if (a.scal) {
            float* sc = a.scal + (size_t)t * NSCAL;
            sc[S_NOISE] = softplus_f(x[0]) + NOISE_LB; sc[S_OS] = softplus_f(x[1]); sc[S_LS] = softplus_f(x[2]);
            const float sn = sigmoid_f(x[0]), ss_ = sigmoid_f(x[1]), sl = sigmoid_f(x[2]);
            sc[S_D1N] = sn; sc[S_D1S] = ss_; sc[S_D1L] = sl;
            sc[S_D2N] = sn * (1.f - sn); sc[S_D2S] = ss_ * (1.f - ss_); sc[S_D2L] = sl * (1.f - sl);
            sc[S_FIN] = f; sc[S_GIN0] = g[0]; sc[S_GIN1] = g[1]; sc[S_GIN2] = g[2];
            sc[S_LOGDET] = extra[0]; sc[S_TRAINV] = extra[1]; sc[S_AA] = extra[2]; sc[S_YA] = extra[3];
            sc[S_TRAINVG] = extra[4]; sc[S_AGA] = extra[5];
            sc[S_GT0] = extra[6]; sc[S_GT1] = extra[7]; sc[S_GT2] = extra[8];
        }
    }
    if (a.vecs && tid < n) a.vecs[((size_t)t * NVEC + V_ALPHA) * a.vld + tid] = sm.alpha[tid];
    if (a.Ainv) {
        float* Ao = a.Ainv + (size_t)t * a.ld * a.ld;
        for (int e = tid; e < n * n; e += NT) {
            const int i = e / n, j = e - i * n;
            Ao[(size_t)i * a.ld + j] = sm.buf0[i * LD + j];
        }
    }
}

}  // namespace adkf
