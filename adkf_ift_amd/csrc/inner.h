// k_inner: one workgroup per task.  Evaluates f_inner = -MLL/N with its exact gradient at phi
// (oracle/closed_form.py::inner_stage) and, in fit mode, runs the whole quasi-Newton inner optimisation
// (the reference's host-side SciPy L-BFGS-B, fs_mol/utils/adaptive_dkt_utils.py:91) inside this one launch.
// Everything big lives in REGISTERS for the entire fit: each thread owns one RB x CB block of the squared
// distances and of the kernel matrix being swept into -(A^-1) (factor.h); LDS carries the pivot rows of the current
// sweep steps, y, alpha, reduction scratch, the optimiser state (FitShared) and one float per matrix element
// (kappa'(u) u, parked by the kernel build for the trace pass): ~14 KB static + NMAX^2 * 4 bytes dynamic.
#pragma once
#include "factor.h"

#ifndef ADKF_EVAL_STAMP
#define ADKF_EVAL_STAMP 0   // diagnostic build only (tools/history/eval_phases.py): s_memtime at the phase boundaries of one evaluation
#endif
#if ADKF_EVAL_STAMP
extern "C" __device__ unsigned long long adkf_eval_stamps[16];
#define ADKF_ES(slot) do { if (blockIdx.x == 8 && threadIdx.x == ADKF_EVAL_STAMP - 1 && adkf_stamp_on) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); adkf_eval_stamps[slot] = t_; } } while (0)
#else
#define ADKF_ES(slot) do {} while (0)
#endif

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

// The squared distances of this thread's block.  LOW = false: register-resident for the whole fit.  LOW = true: parked in the
// kernel's dynamic LDS (lane-private slots, conflict-free) and read back by the kernel build and the trace pass of every
// evaluation, so that the fit lives in <= 128 registers and TWO tasks share a CU (k_inner below): the LDS then carries D^2
// instead of kappa'(u) u, which the trace pass recomputes.
template <int NMAX, int NT, bool LOW = false>
struct D2Block {
    using SW = Sweep<NMAX, NT>;
    static constexpr int RB = SW::RB, CB = SW::CB;
    float reg[LOW ? 1 : RB][LOW ? 1 : CB];
    float* lds;
    __device__ __forceinline__ void init(const float* D2, int ld, int n, float* lds_) {
        const bool vec = rows_aligned16(D2, ld);
        lds = lds_ + threadIdx.x;
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int i = SW::row(r);
            float seg[CB];
            load_segment<CB>(D2 + (size_t)i * ld, SW::col(0), n, i < n, vec, seg);   // exactly symmetric by construction (ProbDist mirrors its tiles); 0 outside n x n
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                if (LOW) lds[(r * CB + c) * NT] = seg[c];
                else reg[LOW ? 0 : r][LOW ? 0 : c] = seg[c];
            }
        }
    }
    __device__ __forceinline__ float get(int r, int c) const { return LOW ? lds[(r * CB + c) * NT] : reg[LOW ? 0 : r][LOW ? 0 : c]; }
};

// f_inner = (nll - log priors) / n and its raw-parameter gradient from the five reductions
// acc = {tr(Ainv G), a^T G a, tr(Ainv), a^T a, y^T a} (oracle/closed_form.py::inner_stage); shared by the register-resident
// evaluator below and the blocked large-N path (large.h).
__device__ __forceinline__ void inner_finalize(int n, const float* x, const float* pri, float logdet, const float* acc,
                                               float& f, float* g, float* extra) {
    const float noise = softplus_f(x[0]) + NOISE_LB, os = softplus_f(x[1]), ls = softplus_f(x[2]);
    const float d1n = sigmoid_f(x[0]), d1s = sigmoid_f(x[1]), d1l = sigmoid_f(x[2]);
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
    extra[0] = logdet; extra[1] = trAinv; extra[2] = aa; extra[3] = ya; extra[4] = trAinvG; extra[5] = aGa;
    extra[6] = gt0; extra[7] = gt1; extra[8] = gt2;
}

// The same, called by ALL 64 lanes of one wave: the transcendental work per raw parameter (softplus, sigmoid, the log-normal
// prior term: ~10 dependent libm calls when one lane does all three) runs on three lanes side by side and is gathered with
// readlane.  Same functions, same order of the two prior terms: the results are bit-identical to inner_finalize.
__device__ __forceinline__ void inner_finalize_wave(int n, const float* x, const float* pri, float logdet, const float* acc,
                                                    float& f, float* g, float* extra) {
    const int lane = threadIdx.x & 63;
    const float xq = lane == 0 ? x[0] : (lane == 1 ? x[1] : x[2]);
    const float tq = softplus_f(xq) + (lane == 0 ? NOISE_LB : 0.f);          // noise | outputscale | lengthscale
    const float dq = sigmoid_f(xq);
    const float pm = lane == 0 ? pri[0] : pri[2], ps = lane == 0 ? pri[1] : (lane == 2 ? pri[3] : -1.f);   // no prior on the outputscale
    float lpq = 0.f, dpq = 0.f;
    if (ps > 0.f) {
        const float lx = logf(tq), z = (lx - pm) / (ps * ps);
        lpq = -lx - logf(ps) - 0.5f * LOG_2PI - 0.5f * (lx - pm) * z;
        dpq = (-1.f - z) / tq;
    }
    auto rl = [](float v, int l) { return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l)); };
    const float noise = rl(tq, 0), os = rl(tq, 1);
    const float d1n = rl(dq, 0), d1s = rl(dq, 1), d1l = rl(dq, 2);
    float lp = 0.f;
    lp += rl(lpq, 0); lp += rl(lpq, 2);
    const float dpn = rl(dpq, 0), dpl = rl(dpq, 2);
    const float trAinvG = acc[0], aGa = acc[1], trAinv = acc[2], aa = acc[3], ya = acc[4];
    const float fn = (float)n;
    const float nll = 0.5f * ya + 0.5f * logdet + 0.5f * fn * LOG_2PI;
    f = (nll - lp) / fn;
    const float gt0 = 0.5f * trAinv - 0.5f * aa - dpn;
    const float gt1 = (0.5f * (fn - noise * trAinv) - 0.5f * (ya - noise * aa)) / os;
    const float gt2 = 0.5f * trAinvG - 0.5f * aGa - dpl;
    g[0] = gt0 * d1n / fn;
    g[1] = gt1 * d1s / fn;
    g[2] = gt2 * d1l / fn;
    extra[0] = logdet; extra[1] = trAinv; extra[2] = aa; extra[3] = ya; extra[4] = trAinvG; extra[5] = aGa;
    extra[6] = gt0; extra[7] = gt1; extra[8] = gt2;
}

template <int NMAX, int NT, int KIND, bool LOW = false>
struct InnerEval {
    using D2 = D2Block<NMAX, NT, LOW>;
    using SW = Sweep<NMAX, NT>;
    static constexpr int RB = SW::RB, CB = SW::CB;

    // One evaluation at raw parameters x.  d2 = this thread's block of squared distances.  On return m = -(A^-1)
    // (this thread's block), sm.vec_out = alpha.  extra (9 floats) receives the scalars later stages reuse.
    // cache: NT * RB * CB floats of LDS; carries kappa'(u) u from the kernel build to the trace pass (no second exp).
    __device__ static __forceinline__ int run(SweepSmem<NMAX, NT>& sm, const D2& d2, float (&m)[RB][CB], int n,
                                              const float* x, const float* tr, const float* pri, float& f, float* g,
                                              float* extra, bool fast, float* cache) {
        int j0 = SW::bc() * CB, i0 = SW::row(0);
        const int tid = threadIdx.x;
        if constexpr (LOW) {
            // (two tasks per CU, <= 128 registers) row and column of this lane's block from copies the optimiser cannot see through:
            // hoisted out of the fit's loop, the sixteen per-row values (row, row - column) of the unrolled passes below stayed alive
            // across the whole fit and were what spilled
            asm volatile("" : "+v"(j0), "+v"(i0));
        }
        auto row_of = [&](int r) { return LOW ? i0 + (r << 4) : SW::row(r); };
#if ADKF_EVAL_STAMP
        const bool adkf_stamp_on = fast;   // the search evaluations (the final one uses libm expf)
#endif
        ADKF_ES(0);
        const float noise = tr[0], os = tr[1], ls = tr[2];   // softplus of x, computed once per trial point (FitShared)
        const float il2 = 1.f / (ls * ls), gl = -2.f / ls;
        if (fast) {
            // Search evaluations: exp as ONE v_exp_f32 of a pre-scaled argument (RBF: d2 * (-log2(e) / (2 l^2)); Matern: -sqrt(5)
            // log2(e) r): 1 ulp of the hardware exp2 plus the rounding of its argument (relative 2^-24 |arg|, i.e. below 2e-6 even
            // where the kernel value is 1e-9) - the noise level of the float32 matrix entries themselves - and no range checks when
            // the task fills the block (every C2 task).  The reported evaluation (fast == false) takes the loop with libm's expf
            // below.  RBF: 6 instead of 22 VALU instructions per element; the loop was 8 % of an evaluation (5.7 k -> 1.4 k cycles).
            const bool full = n == NMAX;   // workgroup-uniform
            const float ce = -0.72134752044448170368f * il2, ck = -0.5f * il2;   // RBF: -log2(e) / (2 l^2);  kappa'(u) u = k0 d2 ck
            const float cm = -3.2259784787f;                                        // Matern: -sqrt(5) log2(e)
#pragma unroll
            for (int r = 0; r < RB; ++r) {
                const int i = row_of(r), dc = i - j0;   // the diagonal sits at column offset dc, if 0 <= dc < CB
#pragma unroll
                for (int c = 0; c < CB; ++c) {
                    if (c % 4 == 0) __builtin_amdgcn_sched_barrier(0);   // four exponentials in flight at a time: keeps the pressure of this loop out of the sweep's allocation
                    const float d2v = d2.get(r, c);
                    float k0, k1u;
                    if (KIND == 0) {
                        k0 = __builtin_amdgcn_exp2f(d2v * ce);
                        k1u = k0 * (d2v * ck);
                    } else {
                        const float u = d2v * il2, sr = SQRT5 * __builtin_amdgcn_sqrtf(u);
                        const float e = __builtin_amdgcn_exp2f(sr * (cm / SQRT5));
                        k0 = (1.f + sr + (5.f / 3.f) * u) * e;
                        k1u = -(5.f / 6.f) * (1.f + sr) * e * u;
                    }
                    float mv = fmaf(os, k0, dc == c ? noise : 0.f);
                    if (!full) {
                        const bool in = i < n && j0 + c < n;
                        mv = in ? mv : (dc == c ? 1.f : 0.f);
                        k1u = in ? k1u : 0.f;
                    }
                    m[r][c] = mv;
                    if (!LOW) cache[(r * CB + c) * NT + tid] = k1u;   // lane-private slots, conflict-free
                }
            }
        } else
#pragma unroll
        for (int r = 0; r < RB; ++r) {
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                if (c % 4 == 0) __builtin_amdgcn_sched_barrier(0);   // four exponentials in flight at a time: keeps the pressure of this loop out of the sweep's allocation
                const int i = row_of(r), j = j0 + c;
                float k1u = 0.f;
                if (i < n && j < n) {
                    const float u = d2.get(r, c) * il2;
                    float k0, k1, k2;
                    if (fast) kappa3<KIND, true>(u, k0, k1, k2); else kappa3<KIND, false>(u, k0, k1, k2);
                    m[r][c] = os * k0 + (i == j ? noise : 0.f);
                    k1u = k1 * u;
                }
                else m[r][c] = (i == j) ? 1.f : 0.f;
                if (!LOW) cache[(r * CB + c) * NT + tid] = k1u;   // lane-private slots, conflict-free
            }
        }
        ADKF_ES(1);
        __syncthreads();  // previous readers of sm (cross/vec_out) are done
        ADKF_ES(10);
        SW::run(m, n, sm);
        ADKF_ES(2);
        SW::solve(m, sm.vec_in, sm.vec_out);  // alpha = A^-1 y
        ADKF_ES(3);
        ADKF_ES(4);
        // tr(Ainv G), a^T G a, tr(Ainv), a^T a, y^T a, log|A|, number of non-positive pivots: ONE block reduction
        float acc[7] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        float ai[RB], aj[CB];
#pragma unroll
        for (int r = 0; r < RB; ++r) ai[r] = sm.vec_out[row_of(r)];
#pragma unroll
        for (int c = 0; c < CB; ++c) aj[c] = sm.vec_out[j0 + c];
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int c = 0; c < CB; ++c) {
                // dK/dl = s kappa'(u) u (-2/l): kappa'(u) u was parked in LDS by the build (zero outside n x n); the
                // two-tasks-per-CU variant keeps D^2 there instead and forms it again, with the build's own arithmetic
                float k1u_;
                if (!LOW) k1u_ = cache[(r * CB + c) * NT + tid];
                else {
                    const float d2v = d2.get(r, c);
                    const bool in = row_of(r) < n && j0 + c < n;
                    if (fast) {
                        // (the build's expressions, letter for letter: the two variants must run the same fit)
                        const float ce = -0.72134752044448170368f * il2, ck = -0.5f * il2, cm = -3.2259784787f;
                        if (KIND == 0) {
                            const float k0 = __builtin_amdgcn_exp2f(d2v * ce);
                            k1u_ = k0 * (d2v * ck);
                        } else {
                            const float u = d2v * il2, sr = SQRT5 * __builtin_amdgcn_sqrtf(u);
                            const float e = __builtin_amdgcn_exp2f(sr * (cm / SQRT5));
                            k1u_ = -(5.f / 6.f) * (1.f + sr) * e * u;
                        }
                    } else {
                        const float u = d2v * il2;
                        float k0, k1, k2;
                        kappa3<KIND, false>(u, k0, k1, k2);
                        k1u_ = k1 * u;
                    }
                    k1u_ = in ? k1u_ : 0.f;
                }
                const float G = os * gl * k1u_;
                acc[0] -= m[r][c] * G;
                acc[1] += ai[r] * aj[c] * G;
                if (row_of(r) == j0 + c && row_of(r) < n) acc[2] -= m[r][c];
            }
        if (tid < n) {
            const float a = sm.vec_out[tid];
            acc[3] = a * a;
            acc[4] = sm.vec_in[tid] * a;
            const float p = sm.pivs[tid];
            acc[5] = logf(p);
            acc[6] = (p > 0.f) ? 0.f : 1.f;
        }
        ADKF_ES(5);
        block_sum<7, NT>(acc, sm.red);
        ADKF_ES(6);
        float logdet = acc[5];
        int info = 0;
        if (acc[6] > 0.f) info = SW::finish(n, sm, logdet);   // rare: locate the first non-positive pivot (uniform branch)
        if (tid >= 64) { f = 0.f; return info; }   // the scalar epilogue is consumed by lane 0 only: one wave computes it
        inner_finalize_wave(n, x, pri, logdet, acc, f, g, extra);
        ADKF_ES(7);
        if (info != 0 || !(f == f)) {
            f = INFINITY;
            return info != 0 ? info : n + 1;
        }
        return 0;
    }
};

// Quasi-Newton driver state (identical in every lane: all inputs come from block-wide reductions).
constexpr float MAX_MOVE = 16.f;

struct Bfgs {
    float x[3], f, g[3];      // current accepted point
    float Hi[3][3];           // inverse-Hessian approximation
    float p[3], gp, step;     // search direction, directional derivative, trial step
    int bt;                   // backtracks on the current direction
    bool first;

    __device__ __forceinline__ void reset_H() {
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) Hi[i][j] = (i == j) ? 1.f : 0.f;
        first = true;
    }
    // new search direction from (g, Hi); false when the gradient vanishes
    __device__ __forceinline__ bool direction() {
#pragma unroll
        for (int i = 0; i < 3; ++i) p[i] = -(Hi[i][0] * g[0] + Hi[i][1] * g[1] + Hi[i][2] * g[2]);
        gp = g[0] * p[0] + g[1] * p[1] + g[2] * p[2];
        if (!(gp < 0.f)) {  // not a descent direction (or NaN): restart from steepest descent
            reset_H();
#pragma unroll
            for (int i = 0; i < 3; ++i) p[i] = -g[i];
            gp = -(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
            if (!(gp < 0.f)) return false;
        }
        step = first ? fminf(1.f, 1.f / (fabsf(g[0]) + fabsf(g[1]) + fabsf(g[2]))) : 1.f;
        // no raw parameter moves by more than MAX_MOVE per trial: a quasi-Newton direction built from differences at
        // the fp32 noise floor can be astronomically long, and far out (outputscale 1e8) the fp32 value is garbage
        // that would pass the Armijo test
        step = fminf(step, MAX_MOVE / fmaxf(fmaxf(fabsf(p[0]), fabsf(p[1])), fmaxf(fabsf(p[2]), 1e-30f)));
        bt = 0;
        return true;
    }
    __device__ __forceinline__ void trial(float* xe) const {
#pragma unroll
        for (int i = 0; i < 3; ++i) xe[i] = x[i] + step * p[i];
    }
    // accept (xn, fn, gn): BFGS update of Hi
    __device__ __forceinline__ void accept(const float* xn, float fn, const float* gn) {
        float s[3], yv[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) { s[i] = xn[i] - x[i]; yv[i] = gn[i] - g[i]; x[i] = xn[i]; g[i] = gn[i]; }
        f = fn;
        const float sy = s[0] * yv[0] + s[1] * yv[1] + s[2] * yv[2];
        const float yy = yv[0] * yv[0] + yv[1] * yv[1] + yv[2] * yv[2];
        const float ss = s[0] * s[0] + s[1] * s[1] + s[2] * s[2];
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
    }
};

// Optimiser state shared by the workgroup.  It lives in LDS (not in every lane's registers: wave-uniform floats
// cannot sit in SGPRs on CDNA, and ~40 of them per lane were pushing the 128-wide instance into scratch); lane 0
// advances it between evaluations, everybody reads the next trial point.
struct FitShared {
    Bfgs st;
    float xe[3];
    float tr[3];      // (noise, outputscale, lengthscale) at xe: computed once by the lane that moves xe, read by everybody
    int phase, evals;
    __device__ __forceinline__ void set_transforms() {
        tr[0] = softplus_f(xe[0]) + NOISE_LB; tr[1] = softplus_f(xe[1]); tr[2] = softplus_f(xe[2]);
    }
};

enum { PH_INIT = 0, PH_SEARCH, PH_BURN, PH_FINAL };

// Consumes the evaluation (fe, ge, ie) made at fs.xe and decides what is evaluated next (lane 0 only).
__device__ __forceinline__ void fit_advance(FitShared& fs, const InnerArgs& a, float fe, const float* ge, int ie, bool transforms = true) {
    Bfgs& st = fs.st;
    const int budget = a.max_evals - 1;  // the last evaluation is the output evaluation
    int phase = fs.phase;
    const int evals = ++fs.evals;
    bool stop = false;
    if (phase == PH_INIT) {
        st.f = fe;
        for (int q = 0; q < 3; ++q) st.g[q] = ge[q];
        if (ie != 0) stop = true;  // infeasible start: reported by the final evaluation
        else if (fmaxf(fabsf(ge[0]), fmaxf(fabsf(ge[1]), fabsf(ge[2]))) <= a.gtol) stop = true;
        else if (!st.direction()) stop = true;
        phase = PH_SEARCH;
    } else if (phase == PH_SEARCH) {
        if (ie == 0 && fe <= st.f + 1e-4f * st.step * st.gp) {  // Armijo (a failed factorisation never counts as progress)
            const float fprev = st.f;
            st.accept(fs.xe, fe, ge);
            const float gmax = fmaxf(fabsf(ge[0]), fmaxf(fabsf(ge[1]), fabsf(ge[2])));
            if (gmax <= a.gtol || fabsf(fprev - fe) <= a.ftol * fmaxf(fmaxf(fabsf(fprev), fabsf(fe)), 1.f)) stop = true;
            else if (!st.direction()) stop = true;
        } else {
            // safeguarded quadratic interpolation of the step
            const float denom = 2.f * (fe - st.f - st.gp * st.step);
            const float sq = (denom > 0.f && fe < INFINITY) ? (-st.gp * st.step * st.step / denom) : 0.5f * st.step;
            st.step = fminf(fmaxf(sq, 0.1f * st.step), 0.5f * st.step);
            if (++st.bt >= 12) stop = true;  // line search failed: converged to working precision
        }
    }
    // exact_evals (benchmark mode): the same stopping rules, but a converged task spends the rest of its budget
    // re-evaluating at its optimum, so every task costs exactly max_evals evaluations and phi* does not depend on the
    // mode.  (Iterating on past convergence instead feeds rounding noise to the quasi-Newton update.)
    if (stop && a.exact_evals) phase = PH_BURN;
    if (evals >= budget || (stop && !a.exact_evals)) {
        phase = PH_FINAL;
        for (int q = 0; q < 3; ++q) fs.xe[q] = st.x[q];
    } else if (phase == PH_BURN) {
        for (int q = 0; q < 3; ++q) fs.xe[q] = st.x[q];
    } else {
        st.trial(fs.xe);
    }
    if (transforms) fs.set_transforms();   // (k_inner computes the three softplus on three lanes instead)
    fs.phase = phase;
}

// max pivot / min pivot over pivs[0..n) (n <= 128: the first two waves hold them); +inf when a pivot is not positive.
// Contains two barriers; all threads call; all get the value.
template <int NT>
__device__ __forceinline__ float pivot_ratio(const float* pivs, int n, float* red) {
    const int tid = threadIdx.x;
    float lo = INFINITY, hi = 0.f;
    for (int k = tid; k < n; k += NT) { const float p = pivs[k]; lo = fminf(lo, p); hi = fmaxf(hi, p); }
    lo = wave_min(lo); hi = wave_max(hi);
    __syncthreads();
    if ((tid & 63) == 0) { red[2 * (tid >> 6)] = lo; red[2 * (tid >> 6) + 1] = hi; }
    __syncthreads();
    for (int w = 0; w < NT / 64; ++w) { lo = fminf(lo, red[2 * w]); hi = fmaxf(hi, red[2 * w + 1]); }
    return lo > 0.f ? hi / lo : INFINITY;
}

// The per-task scalars later stages (Hessian, outer NLL, mixed term) read back from the workspace.
__device__ __forceinline__ void write_inner_scal(float* sc, const float* xe, float f, const float* g, const float* extra) {
    sc[S_NOISE] = softplus_f(xe[0]) + NOISE_LB; sc[S_OS] = softplus_f(xe[1]); sc[S_LS] = softplus_f(xe[2]);
    const float sn = sigmoid_f(xe[0]), ss_ = sigmoid_f(xe[1]), sl = sigmoid_f(xe[2]);
    sc[S_D1N] = sn; sc[S_D1S] = ss_; sc[S_D1L] = sl;
    sc[S_D2N] = sn * (1.f - sn); sc[S_D2S] = ss_ * (1.f - ss_); sc[S_D2L] = sl * (1.f - sl);
    sc[S_FIN] = f; sc[S_GIN0] = g[0]; sc[S_GIN1] = g[1]; sc[S_GIN2] = g[2];
    sc[S_LOGDET] = extra[0]; sc[S_TRAINV] = extra[1]; sc[S_AA] = extra[2]; sc[S_YA] = extra[3];
    sc[S_TRAINVG] = extra[4]; sc[S_AGA] = extra[5];
    sc[S_GT0] = extra[6]; sc[S_GT1] = extra[7]; sc[S_GT2] = extra[8];
    sc[S_AREF] = 0.f;   // a fresh alpha: not refined yet (k_alpha_refine)
}

// LOW = true (128 points only; launch_inner_k picks it when the batch has more tasks than the chip has CUs): the fit in <= 128
// registers - 4 waves per SIMD, i.e. two workgroups per CU - with D^2 in LDS (D2Block above).  A block step of the sweep keeps
// the matrix pipe busy for less than half of its duration (factor_m.h), so a second task on the same CU fills the rest.
template <int NMAX, int NT, int KIND, bool LOW = false>
__global__ __launch_bounds__(NT, LOW ? 4 : 1) void k_inner(InnerArgs a) {
    using EV = InnerEval<NMAX, NT, KIND, LOW>;
    using SW = Sweep<NMAX, NT>;
    constexpr int RB = SW::RB, CB = SW::CB;
    __shared__ SweepSmem<NMAX, NT> sm;
    __shared__ FitShared fs;
    extern __shared__ float inner_cache[];   // NT * RB * CB floats (launch_inner_k sets the dynamic size): kappa'(u) u, or D^2 (LOW)
    int t, tile;
    if (!task_tile(a.T, 1, t, tile)) return;
    const int tid = threadIdx.x;
    const int n = a.n_s ? a.n_s[t] : a.ld;
    const float* D2 = a.D2ss + (size_t)t * a.ld * a.ld;
    const int j0 = SW::bc() * CB;

    typename EV::D2 d2;
    d2.init(D2, a.ld, n, inner_cache);
    float m[RB][CB];
    if (tid < NMAX) sm.vec_in[tid] = (tid < n) ? a.y_s[(size_t)t * a.ld + tid] : 0.f;
    float pri[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) pri[q] = a.priors[t * 4 + q];
    if (tid == 0) {
        for (int q = 0; q < 3; ++q) { fs.st.x[q] = a.phi[t * 3 + q]; fs.xe[q] = fs.st.x[q]; }
        fs.st.reset_H();
        fs.st.f = INFINITY;
        fs.phase = (a.max_evals > 0) ? PH_INIT : PH_FINAL;
        fs.evals = 0;
        fs.set_transforms();
    }
    __syncthreads();

    // One call site of the evaluator; the state machine in fit_advance() picks the next point.
    float xe[3], fe, ge[3], extra[9];
    int info = 0, evals = 0;
    while (true) {
        const int phase = fs.phase;
        xe[0] = fs.xe[0]; xe[1] = fs.xe[1]; xe[2] = fs.xe[2];
        const float tr[3] = {fs.tr[0], fs.tr[1], fs.tr[2]};
        const int ie = EV::run(sm, d2, m, n, xe, tr, pri, fe, ge, extra, phase != PH_FINAL, inner_cache);
        if (phase == PH_FINAL) { info = ie; evals = fs.evals + 1; break; }
        if (tid < 64) {   // wave 0: lane 0 moves the state machine, then lanes 0..2 transform one raw parameter each
            if (tid == 0) fit_advance(fs, a, fe, ge, ie, false);
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");   // (LDS runs one wave's instructions in order: only the compiler needs telling)
            if (tid < 3) fs.tr[tid] = softplus_f(fs.xe[tid]) + (tid == 0 ? NOISE_LB : 0.f);
        }
        __syncthreads();
    }
    const float f = fe;
    const float* g = ge;

    if (tid == 0) {
        if (a.max_evals > 0) { a.phi[t * 3 + 0] = xe[0]; a.phi[t * 3 + 1] = xe[1]; a.phi[t * 3 + 2] = xe[2]; }
        a.info[t] = info;
        if (a.f_out) a.f_out[t] = f;
        if (a.g_out) { a.g_out[t * 3 + 0] = g[0]; a.g_out[t * 3 + 1] = g[1]; a.g_out[t * 3 + 2] = g[2]; }
        if (a.gnorm_out) a.gnorm_out[t] = fmaxf(fabsf(g[0]), fmaxf(fabsf(g[1]), fabsf(g[2])));
        if (a.nevals_out) a.nevals_out[t] = evals;
        if (a.scal) write_inner_scal(a.scal + (size_t)t * NSCAL, xe, f, g, extra);
    }
    if (a.scal) {   // pivot ratio of the final sweep and the largest diagonal entry of A^-1: the cheap condition estimates by
                    // which refine64.h / ProbCres pick the tasks that need more than the explicit float32 inverse
        const float pr = pivot_ratio<NT>(sm.pivs, n, sm.red);
        float dmax = 0.f;
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int c = 0; c < CB; ++c)
                if (SW::row(r) == j0 + c && SW::row(r) < n) dmax = fmaxf(dmax, -m[r][c]);
        dmax = wave_max(dmax);
        __syncthreads();
        if ((tid & 63) == 0) sm.red[tid >> 6] = dmax;
        __syncthreads();
        if (tid == 0) {
            for (int w = 1; w < NT / 64; ++w) dmax = fmaxf(dmax, sm.red[w]);
            float* sc = a.scal + (size_t)t * NSCAL;
            sc[S_PIVR_A] = pr;
            sc[S_CONDA] = (sc[S_OS] + sc[S_NOISE]) * dmax;
        }
    }
    if (a.vecs && tid < n) a.vecs[((size_t)t * NVEC + V_ALPHA) * a.vld + tid] = sm.vec_out[tid];
    if (a.Ainv) {
        // the output addresses are formed HERE from opaque copies: hoisted to the kernel's head (they are loop-invariant) the
        // row offset was the one value (8 bytes) that spilled across the whole fit
        int t_late = t, i_late = SW::row(0), ld_late = a.ld, j_late = j0;
        asm volatile("" : "+s"(t_late), "+s"(ld_late));   // (the leading dimension too: the row offsets i * ld are the ones the loads of D^2 at the kernel's head use)
        asm volatile("" : "+v"(i_late), "+v"(j_late));
        float* Ao = a.Ainv + (size_t)t_late * ld_late * ld_late;
        const bool vec = rows_aligned16(Ao, ld_late);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            const int i = i_late + (SW::row(r) - SW::row(0));   // (a compile-time offset in every layout)
            float neg[CB];
#pragma unroll
            for (int c = 0; c < CB; ++c) neg[c] = -m[r][c];
            store_segment<CB>(Ao + (size_t)i * ld_late, j_late, n, i < n, vec, neg);
        }
    }
}

}  // namespace adkf
