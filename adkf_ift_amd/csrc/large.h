// Support / query sets beyond the register-resident sweep (more than 128 points): the same symmetric sweep, blocked by
// LB = 128 pivots over a matrix that lives in HBM / L2, with the O(N^3) part on the fp32 MFMA (k_bgemm):
//
//   per block step P = [p0, p0 + 128):
//     k_lg_diag     one workgroup per task: D = M_PP into registers, factor.h sweep  ->  Dinv, pivots (log-det, info)
//     ProbLgPanel   F = Dinv * M_P.   (128 x n, K = 128); the epilogue also snapshots C = M_P.
//     ProbLgUpdate  M_RR -= C_R^T F_R (n x n, K = 128);  pivot-row / pivot-column / pivot-block tiles do no product,
//                   their epilogue writes F_R, F_R^T and -Dinv
//   after the last step M = -(A^-1).
// (A look-ahead that factorises the next diagonal block on a side stream while the rest of the update runs was measured:
//  the two event hand-offs per block step cost more than the overlap buys - C5 15.6 -> 17.4 ms, N = 256: 4.8 -> 5.3 ms.
//  Round 4 tried the event-free version - the batch cut into task groups whose launch sequences run on streams of their own, one
//  fork and one join per fit, so that one group's diagonal sweeps sit under another group's update (commit ef34797): the kernels do
//  overlap (profiles/r04_c5_groups.txt) but a task's launches only ever use the 32 CUs of its XCD, so with 8 tasks there is nothing
//  on an XCD to hide behind (C5 13.5 -> 15.0 ms), and from 16 tasks on the gain is 3-5 % (16 x 1024: 18.0 -> 17.2 ms; 32 x 512:
//  7.89 -> 7.63; 64 x 256: 3.60 -> 3.79, slower) - removed again.  What did pay: DEEP and the upper-triangle grid below.)
//
// One MLL evaluation = k_lg_build (kernel matrix from the squared distances) + the block steps + k_lg_matvec (alpha) +
// k_lg_traces (the three O(N^2) reductions; also flips the sign in place) + k_lg_advance (value, gradient, and one
// transition of the SAME BFGS/Armijo state machine k_inner runs, fit_advance(), on state kept in the workspace).
// The host enqueues max_evals evaluations back to back without synchronising; tasks that have finished are skipped by
// every kernel (phase == PH_DONE).  The outer factorisation of S reuses the block steps.
#pragma once
#include "kernels.h"

namespace adkf {

constexpr int LB = 128;
enum { PH_DONE = 4 };

// The matrix being swept in place (per task) and the side buffers of one block step.
struct LgMat {
    float* M; int ld; const int32_t* n_arr;
    const FitShared* fit;   // null: every task is active
    float* Dinv;            // [T, LB, LB]
    float* Cbuf;            // [T, LB, ld]
    float* Fbuf;            // [T, LB, ld]
    float* logdet;          // [T] running log-determinant
    float* pext;            // [T, 2] smallest / largest pivot so far: the condition estimate that picks the float64 path (refine64.h)
    int32_t* info;          // [T] first non-positive pivot (1-based) or 0
    int32_t* cnt;           // [T] arrival counters of the fused block step (large_fused.h); null on the three-launch path
    int T; bool vec;
    __device__ __forceinline__ bool active(int t) const { return !fit || fit[t].phase != PH_DONE; }
    __device__ __forceinline__ int n(int t) const { return n_arr ? n_arr[t] : ld; }
};

__global__ __launch_bounds__(512) void k_lg_diag(LgMat a, int step) {
    using SW = Sweep<128, 512>;
    constexpr int RB = SW::RB, CB = SW::CB;
    __shared__ SweepSmem<128, 512> sm;
    int t, tile;
    if (!task_tile(a.T, 1, t, tile)) return;
    if (!a.active(t)) return;
    const int n = a.n(t), p0 = step * LB;
    const int nloc = min(LB, n - p0);
    if (nloc <= 0) return;
    const float* Mi = a.M + (size_t)t * a.ld * a.ld;
    // the diagonal block, identity-padded; M is exactly symmetric (ProbLgUpdate writes every tile together with its mirror image),
    // so the rows are read as they are: 8 sixteen-byte loads per lane instead of 32 conditional dword loads
    float m[RB][CB];
    const float* blk = Mi + (size_t)p0 * a.ld + p0;
    const bool vec = rows_aligned16(blk, a.ld);
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int i = SW::row(r);
        load_segment<CB>(blk + (size_t)i * a.ld, SW::col(0), nloc, i < nloc, vec, m[r]);
#pragma unroll
        for (int c = 0; c < CB; ++c)
            if (i == SW::col(c) && i >= nloc) m[r][c] = 1.f;
    }
    __syncthreads();
    SW::run(m, nloc, sm);
    float logdet;
    const int info = SW::finish(nloc, sm, logdet);
    float* Dv = a.Dinv + (size_t)t * LB * LB;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        float neg[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c) neg[c] = -m[r][c];
        store_segment<CB>(Dv + SW::row(r) * LB, SW::col(0), LB, true, true, neg);   // Dinv is [LB, LB] in the 256-byte-aligned workspace
    }
    // extreme pivots of this block (rows beyond nloc are identity padding: pivot 1, not counted)
    float plo = INFINITY, phi = 0.f;
    if ((int)threadIdx.x < nloc) { plo = phi = sm.pivs[threadIdx.x]; }
    plo = wave_min(plo); phi = wave_max(phi);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sm.red[2 * (threadIdx.x >> 6)] = plo; sm.red[2 * (threadIdx.x >> 6) + 1] = phi; }
    __syncthreads();
    if (threadIdx.x == 0) {
        if (a.cnt && step == 0) a.cnt[t] = 0;   // (a sweep starts here: the counters of large_fused.h are zero from now on)
        a.logdet[t] = (step == 0 ? 0.f : a.logdet[t]) + logdet;
        const int prev = step == 0 ? 0 : a.info[t];
        a.info[t] = prev != 0 ? prev : (info != 0 ? p0 + info : 0);
        plo = fminf(sm.red[0], sm.red[2]); phi = fmaxf(sm.red[1], sm.red[3]);      // the pivots sit in the first two waves
        if (a.pext) {
            a.pext[2 * t] = step == 0 ? plo : fminf(a.pext[2 * t], plo);
            a.pext[2 * t + 1] = step == 0 ? phi : fmaxf(a.pext[2 * t + 1], phi);
        }
    }
}

struct ProbLgPanel {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = false;
    static constexpr int NRED = 0;
    LgMat m; int step;
    int n, p0, nloc; const float *Dv, *Mi; float *Cb, *Fb; bool vec;
    __device__ bool setup(int t) {
        if (!m.active(t)) return false;
        n = m.n(t); p0 = step * LB; nloc = min(LB, n - p0); vec = m.vec;
        if (nloc <= 0) return false;
        Dv = m.Dinv + (size_t)t * LB * LB; Mi = m.M + (size_t)t * m.ld * m.ld;
        Cb = m.Cbuf + (size_t)t * LB * m.ld; Fb = m.Fbuf + (size_t)t * LB * m.ld;
        return true;
    }
    __device__ int M() const { return nloc; } __device__ int N() const { return n; } __device__ int K() const { return nloc; }
    __device__ bool skip(int, int n0) const { return n0 >= p0 && n0 < p0 + LB; }  // F_P is never read
    __device__ float a(int i, int k) const { return Dv[i * LB + k]; }
    __device__ float b(int k, int j) const { return Mi[(size_t)(p0 + k) * m.ld + j]; }
    __device__ void a4(int i, int k, float (&v)[4]) const { ld4(Dv + i * LB + k, v); }
    __device__ void b4(int k, int j, float (&v)[4]) const { ld4(Mi + (size_t)(p0 + k) * m.ld + j, v); }
    static constexpr int A_NRAW = 1, B_NRAW = 1;   // two-phase operand path of gemm.h
    __device__ bool raw_ok() const { return true; }
    __device__ void a_raw(int i, int k, float4 (&r)[1]) const { r[0] = ldq(Dv + i * LB + k); }
    __device__ void a_fin(int, int, const float4 (&r)[1], float (&v)[4]) const { unq(r[0], v); }
    __device__ void b_raw(int k, int j, float4 (&r)[1]) const { r[0] = ldq(Mi + (size_t)(p0 + k) * m.ld + j); }
    __device__ void b_fin(int, int, const float4 (&r)[1], float (&v)[4]) const { unq(r[0], v); }
    __device__ void epi(int i, int j, float acc, float*) const {
        Fb[(size_t)i * m.ld + j] = acc;
        Cb[(size_t)i * m.ld + j] = Mi[(size_t)(p0 + i) * m.ld + j];
    }
    // full block steps (K = 128 = DEEP chunks): all operand loads and the snapshot's source in flight before the first MFMA (gemm.h)
    static constexpr int DEEP = LB / GK;
    __device__ void pre4(int i0, int j, float (&v)[4]) const {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = Mi[(size_t)(p0 + i0 + r) * m.ld + j];
    }
    __device__ void epi4(int i0, int j, const float (&acc)[4], float* red) const {
#pragma unroll
        for (int r = 0; r < 4; ++r) epi(i0 + r, j, acc[r], red);
    }
    __device__ void epi4p(int i0, int j, const float (&acc)[4], const float (&pre)[4], float*) const {
#pragma unroll
        for (int r = 0; r < 4; ++r) { Fb[(size_t)(i0 + r) * m.ld + j] = acc[r]; Cb[(size_t)(i0 + r) * m.ld + j] = pre[r]; }
    }
    __device__ void store_red(int, const float*) const {}
};

struct ProbLgUpdate {
    static constexpr bool A_KCONTIG = false, B_KCONTIG = false;
    static constexpr int NRED = 0;
    LgMat m; int step;
    int n, p0, nloc; const float *Dv, *Cb, *Fb; float* Mi; bool vec;
    int tri;   // > 0: the launch has tri = tn (tn + 1) / 2 workgroups per task, one per tile on or above the diagonal (the others would exit at once)
    __device__ bool map(int tiles, int& t, int& tile) const {
        if (tri <= 0) return task_tile(m.T, tiles, t, tile);
        int u;
        if (!task_tile(m.T, tri, t, u)) return false;
        // row ti of the upper triangle starts at ti tn - ti (ti - 1) / 2
        const int tn = (int)((sqrtf(8.f * (float)tri + 1.f) - 1.f) * 0.5f + 0.5f);
        int ti = (int)(((float)(2 * tn + 1) - sqrtf((float)((2 * tn + 1) * (2 * tn + 1) - 8 * u))) * 0.5f);
        ti = max(0, min(ti, tn - 1));
        while (ti > 0 && ti * tn - ti * (ti - 1) / 2 > u) --ti;
        while (ti + 1 < tn && (ti + 1) * tn - (ti + 1) * ti / 2 <= u) ++ti;
        tile = ti * tn + ti + (u - (ti * tn - ti * (ti - 1) / 2));
        return true;
    }
    __device__ bool setup(int t) {
        if (!m.active(t)) return false;
        n = m.n(t); p0 = step * LB; nloc = min(LB, n - p0); vec = m.vec;
        if (nloc <= 0) return false;
        Dv = m.Dinv + (size_t)t * LB * LB; Mi = m.M + (size_t)t * m.ld * m.ld;
        Cb = m.Cbuf + (size_t)t * LB * m.ld; Fb = m.Fbuf + (size_t)t * LB * m.ld;
        return true;
    }
    __device__ int M() const { return n; } __device__ int N() const { return n; } __device__ int K() const { return nloc; }
    __device__ bool in_p(int i) const { return i >= p0 && i < p0 + LB; }
    // M stays symmetric: only tiles on or above the diagonal are computed, their epilogue also writes the mirror image
    __device__ bool skip(int m0, int n0) const { return in_p(m0) || in_p(n0) || m0 > n0; }
    __device__ float a(int i, int k) const { return Cb[(size_t)k * m.ld + i]; }
    __device__ float b(int k, int j) const { return Fb[(size_t)k * m.ld + j]; }
    __device__ void a4(int i, int k, float (&v)[4]) const { ld4(Cb + (size_t)k * m.ld + i, v); }
    __device__ void b4(int k, int j, float (&v)[4]) const { ld4(Fb + (size_t)k * m.ld + j, v); }
    static constexpr int A_NRAW = 1, B_NRAW = 1;   // two-phase operand path of gemm.h
    __device__ bool raw_ok() const { return true; }
    __device__ void a_raw(int i, int k, float4 (&r)[1]) const { r[0] = ldq(Cb + (size_t)k * m.ld + i); }
    __device__ void a_fin(int, int, const float4 (&r)[1], float (&v)[4]) const { unq(r[0], v); }
    __device__ void b_raw(int k, int j, float4 (&r)[1]) const { r[0] = ldq(Fb + (size_t)k * m.ld + j); }
    __device__ void b_fin(int, int, const float4 (&r)[1], float (&v)[4]) const { unq(r[0], v); }
    __device__ void epi(int i, int j, float acc, float*) const {
        const int ti = i >> 6, tj = j >> 6;
        if (ti > tj) return;
        float* dst = Mi + (size_t)i * m.ld + j;
        const bool pi = in_p(i), pj = in_p(j);
        float v;
        if (!pi && !pj) v = *dst - acc;
        else if (pi && pj) v = -Dv[(i - p0) * LB + (j - p0)];
        else if (pi) v = Fb[(size_t)(i - p0) * m.ld + j];
        else v = Fb[(size_t)(j - p0) * m.ld + i];
        *dst = v;
        if (ti < tj) Mi[(size_t)j * m.ld + i] = v;
    }
    // four rows of one column: the mirror image is ONE 16-byte store (i0 is a multiple of 4, ld too when vec)
    __device__ void epi4(int i0, int j, const float (&acc)[4], float* red) const {
        const int ti = i0 >> 6, tj = j >> 6;
        if (ti > tj) return;
        if (!vec || ti == tj) {
#pragma unroll
            for (int r = 0; r < 4; ++r) epi(i0 + r, j, acc[r], red);
            return;
        }
        const bool pi = in_p(i0), pj = in_p(j);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* dst = Mi + (size_t)(i0 + r) * m.ld + j;
            if (!pi && !pj) v[r] = *dst - acc[r];
            else if (pi && pj) v[r] = -Dv[(i0 + r - p0) * LB + (j - p0)];
            else if (pi) v[r] = Fb[(size_t)(i0 + r - p0) * m.ld + j];
            else v[r] = Fb[(size_t)(j - p0) * m.ld + i0 + r];
            *dst = v[r];
        }
        *reinterpret_cast<float4*>(Mi + (size_t)j * m.ld + i0) = make_float4(v[0], v[1], v[2], v[3]);
    }
    // a DEEP tile is a computed one: above or on the diagonal, outside the pivot rows and columns
    static constexpr int DEEP = LB / GK;
    __device__ void pre4(int i0, int j, float (&v)[4]) const {
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = Mi[(size_t)(i0 + r) * m.ld + j];
    }
    __device__ void epi4p(int i0, int j, const float (&acc)[4], const float (&pre)[4], float*) const {
        const bool mirror = (i0 >> 6) < (j >> 6);
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = pre[r] - acc[r]; Mi[(size_t)(i0 + r) * m.ld + j] = v[r]; }
        if (mirror) *reinterpret_cast<float4*>(Mi + (size_t)j * m.ld + i0) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __device__ void store_red(int, const float*) const {}
};

// out[i] = sign * sum_j M_ij x_j   (wave per row; grid: ceil(ld / 4) x T)
struct LgMatvecArgs { LgMat m; const float* x; size_t x_stride; float* out; size_t out_stride; float sign; };

__global__ __launch_bounds__(256) void k_lg_matvec(LgMatvecArgs a) {
    const int t = blockIdx.y;
    if (!a.m.active(t)) return;
    const int n = a.m.n(t), i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const float* row = a.m.M + ((size_t)t * a.m.ld + i) * a.m.ld;
    const float* x = a.x + (size_t)t * a.x_stride;
    float s = 0.f;
    for (int j = lane; j < n; j += 64) s += row[j] * x[j];
    s = wave_sum(s);
    if (lane == 0) a.out[(size_t)t * a.out_stride + i] = a.sign * s;
}

// ---- inner (support) side ------------------------------------------------------------------------------------
struct LgInner {
    InnerArgs in;       // the same argument block k_inner takes
    LgMat mat;          // mat.M = in.Ainv
    FitShared* fit;     // [T]
    float* part;        // [T, ntiles, 4] partial reductions of k_lg_traces
    int ntiles, tiles_1d;
};


__global__ void k_lg_begin(LgInner a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.in.T) return;
    FitShared& fs = a.fit[t];
    for (int q = 0; q < 3; ++q) { fs.st.x[q] = a.in.phi[t * 3 + q]; fs.xe[q] = fs.st.x[q]; }
    fs.st.reset_H();
    fs.st.f = INFINITY;
    fs.phase = (a.in.max_evals > 0) ? PH_INIT : PH_FINAL;
    fs.evals = 0;
}

// M = s kappa(D2 / l^2) + noise I at the trial point of the task's state machine (64 x 64 tile per workgroup)
__global__ __launch_bounds__(256) void k_lg_build(LgInner a) {
    int t, tile;
    if (!task_tile(a.in.T, a.ntiles, t, tile)) return;
    if (!a.mat.active(t)) return;
    const int n = a.mat.n(t), ld = a.in.ld;
    const int m0 = (tile / a.tiles_1d) * GT, n0 = (tile % a.tiles_1d) * GT;
    if (m0 >= n || n0 >= n) return;
    const FitShared& fs = a.fit[t];
    const float noise = softplus_f(fs.xe[0]) + NOISE_LB, os = softplus_f(fs.xe[1]), ls = softplus_f(fs.xe[2]);
    const float il2 = 1.f / (ls * ls);
    const float* D2 = a.in.D2ss + (size_t)t * ld * ld;
    float* Mi = a.mat.M + (size_t)t * ld * ld;
    // sixteen elements per thread: all their loads first, to clamped addresses (no branch around a load: kernels.h), then the
    // arithmetic and the predicated stores
    constexpr int EPT = GT * GT / 256;
    const int j = n0 + (threadIdx.x & 63), jc = min(j, n - 1);
    float d2v[EPT];
#pragma unroll
    for (int q = 0; q < EPT; ++q) d2v[q] = D2[(size_t)min(m0 + (threadIdx.x >> 6) + 4 * q, n - 1) * ld + jc];
#pragma unroll
    for (int q = 0; q < EPT; ++q) {
        const int i = m0 + (threadIdx.x >> 6) + 4 * q;
        if (i < n && j < n) Mi[(size_t)i * ld + j] = os * kappa0(a.in.kind, d2v[q] * il2) + (i == j ? noise : 0.f);
    }
}

// After the sweep M = -(A^-1): flip the sign in place and reduce tr(Ainv G), alpha^T G alpha, tr(Ainv) per tile.
__global__ __launch_bounds__(256) void k_lg_traces(LgInner a) {
    __shared__ float red[3 * 4];
    __shared__ float redmax[4];
    int t, tile;
    if (!task_tile(a.in.T, a.ntiles, t, tile)) return;
    if (!a.mat.active(t)) return;
    // (round 5) only the FINAL evaluation has to leave +A^-1 behind: a search evaluation's matrix is overwritten by the next one
    // (k_lg_advance, the next launch, is what moves the phase on: every tile of this launch reads the same value)
    const bool flip = a.fit[t].phase == PH_FINAL;
    const int n = a.mat.n(t), ld = a.in.ld;
    const int m0 = (tile / a.tiles_1d) * GT, n0 = (tile % a.tiles_1d) * GT;
    float acc[3] = {0.f, 0.f, 0.f};
    float dmax = 0.f;   // largest diagonal entry of A^-1 in this tile: with (s + noise) it is the estimate that picks the float32 refinement of C and alpha
    if (m0 < n && n0 < n) {
        const FitShared& fs = a.fit[t];
        const float os = softplus_f(fs.xe[1]), ls = softplus_f(fs.xe[2]);
        const float il2 = 1.f / (ls * ls), gl = -2.f / ls;
        const float* D2 = a.in.D2ss + (size_t)t * ld * ld;
        float* Mi = a.mat.M + (size_t)t * ld * ld;
        const float* al = a.in.vecs + ((size_t)t * NVEC + V_ALPHA) * a.in.vld;
        constexpr int EPT = GT * GT / 256;   // loads first, clamped (see k_lg_build)
        const int j = n0 + (threadIdx.x & 63), jc = min(j, n - 1);
        float mv[EPT], d2v[EPT], aa[EPT];
        const float alj = al[jc];
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int ic = min(m0 + (threadIdx.x >> 6) + 4 * q, n - 1);
            mv[q] = Mi[(size_t)ic * ld + jc]; d2v[q] = D2[(size_t)ic * ld + jc]; aa[q] = al[ic] * alj;
        }
#pragma unroll
        for (int q = 0; q < EPT; ++q) {
            const int i = m0 + (threadIdx.x >> 6) + 4 * q;
            if (i < n && j < n) {
                const float ai = -mv[q];
                if (flip) Mi[(size_t)i * ld + j] = ai;
                float k0, k1, k2;
                const float u = d2v[q] * il2;
                kappa3(a.in.kind, u, k0, k1, k2);
                const float G = os * k1 * u * gl;
                acc[0] += ai * G;
                acc[1] += aa[q] * G;
                if (i == j) { acc[2] += ai; dmax = fmaxf(dmax, ai); }
            }
        }
    }
    block_sum<3, 256>(acc, red);
    dmax = wave_max(dmax);
    if ((threadIdx.x & 63) == 0) redmax[threadIdx.x >> 6] = dmax;
    __syncthreads();
    if (threadIdx.x == 0) {
        float* p = a.part + ((size_t)t * a.ntiles + tile) * 4;
        p[0] = acc[0]; p[1] = acc[1]; p[2] = acc[2];
        p[3] = fmaxf(fmaxf(redmax[0], redmax[1]), fmaxf(redmax[2], redmax[3]));
    }
}

// NT threads per task (round 5: 256 instead of one wave walking the partials and the vectors - a chain of twenty dependent trips to
// memory): finish the evaluation, then either advance the optimiser or publish the final results.  The sums run over fixed index
// sets in a fixed order: deterministic.  (Measured and dropped in round 5: the last trace tile of a task running this step itself
// through an arrival counter - 41 us for the fused launch against 22 + 9 for the two; and block step 0 generating the matrix from
// the squared distances instead of k_lg_build - the exponentials cost the three kernels of that step what the build launch costs.)
template <int NT>
__device__ __forceinline__ void lg_advance_task(const LgInner& a, int t, int tid, float* red) {
    FitShared& fs = a.fit[t];
    if (fs.phase == PH_DONE) return;
    const int n = a.mat.n(t), lane = tid;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    float dmax = 0.f;
    for (int q = tid; q < a.ntiles; q += NT) {
        const float* p = a.part + ((size_t)t * a.ntiles + q) * 4;
        acc[0] += p[0]; acc[1] += p[1]; acc[2] += p[2];
        dmax = fmaxf(dmax, p[3]);
    }
    const float* al = a.in.vecs + ((size_t)t * NVEC + V_ALPHA) * a.in.vld;
    const float* y = a.in.y_s + (size_t)t * a.in.ld;
    for (int i = tid; i < n; i += NT) { const float v = al[i]; acc[3] += v * v; acc[4] += y[i] * v; }
    dmax = wave_max(dmax);
#pragma unroll
    for (int q = 0; q < 5; ++q) acc[q] = wave_sum(acc[q]);
    if constexpr (NT > 64) {
        // per-wave totals through LDS, summed in wave order by everybody
        if ((tid & 63) == 0) {
#pragma unroll
            for (int q = 0; q < 5; ++q) red[(tid >> 6) * 8 + q] = acc[q];
            red[(tid >> 6) * 8 + 5] = dmax;
        }
        __syncthreads();
#pragma unroll
        for (int q = 0; q < 5; ++q) acc[q] = 0.f;
        dmax = 0.f;
        for (int w = 0; w < NT / 64; ++w) {
#pragma unroll
            for (int q = 0; q < 5; ++q) acc[q] += red[w * 8 + q];
            dmax = fmaxf(dmax, red[w * 8 + 5]);
        }
    }
    if (lane != 0) return;
    float xe[3] = {fs.xe[0], fs.xe[1], fs.xe[2]}, pri[4], f, g[3], extra[9];
    for (int q = 0; q < 4; ++q) pri[q] = a.in.priors[t * 4 + q];
    inner_finalize(n, xe, pri, a.mat.logdet[t], acc, f, g, extra);
    int ie = a.mat.info[t];
    if (ie != 0 || !(f == f)) { f = INFINITY; ie = ie != 0 ? ie : n + 1; }
    if (fs.phase != PH_FINAL) { fit_advance(fs, a.in, f, g, ie); return; }
    const InnerArgs& o = a.in;
    if (o.max_evals > 0) { o.phi[t * 3 + 0] = xe[0]; o.phi[t * 3 + 1] = xe[1]; o.phi[t * 3 + 2] = xe[2]; }
    o.info[t] = ie;
    if (o.f_out) o.f_out[t] = f;
    if (o.g_out) { o.g_out[t * 3 + 0] = g[0]; o.g_out[t * 3 + 1] = g[1]; o.g_out[t * 3 + 2] = g[2]; }
    if (o.gnorm_out) o.gnorm_out[t] = fmaxf(fabsf(g[0]), fmaxf(fabsf(g[1]), fabsf(g[2])));
    if (o.nevals_out) o.nevals_out[t] = fs.evals + 1;
    if (o.scal) {
        float* sc = o.scal + (size_t)t * NSCAL;
        write_inner_scal(sc, xe, f, g, extra);
        // the condition estimates of the register path (inner.h), from the blocked sweep: pivot ratio and (s + noise) max diag(A^-1)
        const float plo = a.mat.pext ? a.mat.pext[2 * t] : 0.f, phi = a.mat.pext ? a.mat.pext[2 * t + 1] : 0.f;
        sc[S_PIVR_A] = plo > 0.f ? phi / plo : INFINITY;
        sc[S_CONDA] = (sc[S_OS] + sc[S_NOISE]) * dmax;
    }
    fs.phase = PH_DONE;
}

__global__ __launch_bounds__(256) void k_lg_advance(LgInner a) {
    __shared__ float adv_red[8 * 4];
    if ((int)blockIdx.x >= a.in.T) return;
    lg_advance_task<256>(a, blockIdx.x, threadIdx.x, adv_red);
}

// ---- outer (query) side --------------------------------------------------------------------------------------
// mu = C y_s, r = y_q - mu  (wave per row; grid: ceil(nq_ld / 4) x T)
__global__ __launch_bounds__(256) void k_lg_resid(OuterArgs a) {
    const int t = blockIdx.y;
    const int n = a.tv.ns(t), m = a.tv.nq(t), i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= m) return;
    const float* row = a.C + ((size_t)t * a.tv.nq_ld + i) * a.tv.ns_ld;
    const float* ys = a.y_s + (size_t)t * a.tv.ns_ld;
    float s = 0.f;
    for (int j = lane; j < n; j += 64) s += row[j] * ys[j];
    s = wave_sum(s);
    if (lane == 0) {
        float* vb = a.vecs + (size_t)t * NVEC * a.tv.vld;
        vb[V_MU * a.tv.vld + i] = s;
        vb[V_R * a.tv.vld + i] = a.y_q[(size_t)t * a.tv.nq_ld + i] - s;
    }
}

// S := -S in place (64 x 64 tile per workgroup)
__global__ __launch_bounds__(256) void k_lg_negate(LgMat a, int tiles_1d) {
    int t, tile;
    if (!task_tile(a.T, tiles_1d * tiles_1d, t, tile)) return;
    const int n = a.n(t);
    const int m0 = (tile / tiles_1d) * GT, n0 = (tile % tiles_1d) * GT;
    if (m0 >= n || n0 >= n) return;
    float* Mi = a.M + (size_t)t * a.ld * a.ld;
    for (int e = threadIdx.x; e < GT * GT; e += 256) {
        const int i = m0 + (e >> 6), j = n0 + (e & 63);
        if (i < n && j < n) Mi[(size_t)i * a.ld + j] = -Mi[(size_t)i * a.ld + j];
    }
}

// out[j] = sum_i w_i M_ij (w = null: plain column sums) for an m x n matrix: 64 columns x 16 row groups per workgroup
// (grid: ceil(n / 64) x T), coalesced 256-byte row segments.
struct LgColsumArgs { const float* M; int ld; size_t m_stride; const int32_t* m_arr; int m_ld; const int32_t* n_arr; int n_ld;
                      const float* w; size_t w_stride; float* out; size_t out_stride; };

__global__ __launch_bounds__(1024) void k_lg_colsum(LgColsumArgs a) {
    __shared__ float part[16][64];
    const int t = blockIdx.y, cl = threadIdx.x & 63, g = threadIdx.x >> 6, j = blockIdx.x * 64 + cl;
    const int m = a.m_arr ? a.m_arr[t] : a.m_ld, n = a.n_arr ? a.n_arr[t] : a.n_ld;
    const float* Mi = a.M + (size_t)t * a.m_stride;
    const float* w = a.w ? a.w + (size_t)t * a.w_stride : nullptr;
    float s = 0.f;
    if (j < n)
        for (int i = g; i < m; i += 16) s += (w ? w[i] : 1.f) * Mi[(size_t)i * a.ld + j];
    part[g][cl] = s;
    __syncthreads();
    if (g == 0 && j < n) {
        s = 0.f;
#pragma unroll
        for (int q = 0; q < 16; ++q) s += part[q][cl];
        a.out[(size_t)t * a.out_stride + j] = s;
    }
}

// f_out = (r^T e + log|S| + m log 2 pi) / 2  (one wave per task; Cte = C^T e comes from k_lg_colsum)
struct LgOuterFin { OuterArgs o; const float* logdet; const int32_t* info_s; const float* pext; };

__global__ __launch_bounds__(64) void k_lg_outer_fin(LgOuterFin a) {
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= a.o.T) return;
    const TaskView& tv = a.o.tv;
    const int m = tv.nq(t);
    const float* vb = a.o.vecs + (size_t)t * NVEC * tv.vld;
    float q = 0.f;
    for (int i = lane; i < m; i += 64) q += vb[V_R * tv.vld + i] * vb[V_E * tv.vld + i];
    q = wave_sum(q);
    if (lane == 0) {
        const float logdet = a.logdet[t];
        const int info = a.info_s[t];
        const float f = 0.5f * q + 0.5f * logdet + 0.5f * (float)m * LOG_2PI;
        a.o.scal[(size_t)t * NSCAL + S_FOUT] = f;
        a.o.scal[(size_t)t * NSCAL + S_LOGDETS] = logdet;
        a.o.scal[(size_t)t * NSCAL + S_PIVR_S] = (a.pext && a.pext[2 * t] > 0.f) ? a.pext[2 * t + 1] / a.pext[2 * t] : INFINITY;
        if (a.o.f_out) a.o.f_out[t] = (info == 0) ? f : NAN;
        if (a.o.reset_info) a.o.info[t] = info != 0 ? 100000 + info : 0;
        else if (info != 0 && a.o.info[t] == 0) a.o.info[t] = 100000 + info;
    }
}

// ---- tile-parallel twins of the one-workgroup-per-task kernels of kernels.h (few tasks, many points) ---------------
// k_hess, part 1: beta = G alpha, gamma = Ainv alpha (wave per row; grid: ceil(ns_ld / 4) x T)
__global__ __launch_bounds__(256) void k_lg_hess_mv(HessArgs a) {
    const int t = blockIdx.y, i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    const int n = a.tv.ns(t), ld = a.tv.ns_ld;
    if (i >= n) return;
    const float* sc = a.scal + (size_t)t * NSCAL;
    const float os = sc[S_OS], ls = sc[S_LS], il2 = 1.f / (ls * ls);
    const float* Ai = a.Ainv + ((size_t)t * ld + i) * ld;
    const float* D2 = a.D2ss + ((size_t)t * ld + i) * ld;
    float* vb = a.vecs + (size_t)t * NVEC * a.tv.vld;
    const float* al = vb + V_ALPHA * a.tv.vld;
    float sb = 0.f, sg = 0.f;
    for (int j = lane; j < n; j += 64) {
        float k0, k1, k2; const float u = D2[j] * il2; kappa3(a.tv.kind, u, k0, k1, k2);
        const float aj = al[j];
        sb += os * k1 * u * (-2.f / ls) * aj;
        sg += Ai[j] * aj;
    }
    sb = wave_sum(sb); sg = wave_sum(sg);
    if (lane == 0) { vb[V_BETA * a.tv.vld + i] = sb; vb[V_GAMMA * a.tv.vld + i] = sg; }
}

// k_hess, part 2 (after delta = Ainv beta): the five O(N^2) traces per 64 x 64 tile
struct LgHessTr { HessArgs h; float* part; int ntiles, tiles_1d; };

__global__ __launch_bounds__(256) void k_lg_hess_tr(LgHessTr a) {
    __shared__ float red[5 * 4];
    int t, tile;
    if (!task_tile(a.h.T, a.ntiles, t, tile)) return;
    const int n = a.h.tv.ns(t), ld = a.h.tv.ns_ld;
    const int m0 = (tile / a.tiles_1d) * GT, n0 = (tile % a.tiles_1d) * GT;
    float acc[5] = {0.f, 0.f, 0.f, 0.f, 0.f};
    if (m0 < n && n0 < n) {
        const float* sc = a.h.scal + (size_t)t * NSCAL;
        const float os = sc[S_OS], ls = sc[S_LS], il2 = 1.f / (ls * ls);
        const float* Ai = a.h.Ainv + (size_t)t * ld * ld;
        const float* Pi = a.h.P + (size_t)t * ld * ld;
        const float* D2 = a.h.D2ss + (size_t)t * ld * ld;
        const float* al = a.h.vecs + ((size_t)t * NVEC + V_ALPHA) * a.h.tv.vld;
        for (int e = threadIdx.x; e < GT * GT; e += 256) {
            const int i = m0 + (e >> 6), j = n0 + (e & 63);
            if (i < n && j < n) {
                const float ai = Ai[(size_t)i * ld + j], pij = Pi[(size_t)i * ld + j], pji = Pi[(size_t)j * ld + i];
                float k0, k1, k2; const float u = D2[(size_t)i * ld + j] * il2; kappa3(a.h.tv.kind, u, k0, k1, k2);
                const float Kll = os * (k2 * 4.f * u * u + k1 * 6.f * u) * il2;
                acc[0] += ai * ai; acc[1] += pij * ai; acc[2] += pij * pji; acc[3] += ai * Kll; acc[4] += al[i] * al[j] * Kll;
            }
        }
    }
    block_sum<5, 256>(acc, red);
    if (threadIdx.x == 0) {
        float* p = a.part + ((size_t)t * a.ntiles + tile) * 8;
#pragma unroll
        for (int q = 0; q < 5; ++q) p[q] = acc[q];
    }
}

// k_hess, part 3: sum the partials, the four vector dot products, assemble H (one wave per task)
__global__ __launch_bounds__(64) void k_lg_hess_fin(LgHessTr a) {
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= a.h.T) return;
    const int n = a.h.tv.ns(t), vld = a.h.tv.vld;
    float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int q = lane; q < a.ntiles; q += 64) {
        const float* p = a.part + ((size_t)t * a.ntiles + q) * 8;
#pragma unroll
        for (int k = 0; k < 5; ++k) acc[k] += p[k];
    }
    const float* vb = a.h.vecs + (size_t)t * NVEC * vld;
    for (int i = lane; i < n; i += 64) {
        const float al = vb[V_ALPHA * vld + i], be = vb[V_BETA * vld + i], ga = vb[V_GAMMA * vld + i], de = vb[V_DELTA * vld + i];
        acc[5] += al * ga; acc[6] += be * ga; acc[7] += be * de; acc[8] += al * be;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) acc[k] = wave_sum(acc[k]);
    if (lane == 0) hess_assemble(a.h.scal + (size_t)t * NSCAL, a.h.priors + t * 4, n, acc);
}

// k_wqq per tile; k_lg_wqq_fin sums the three reductions into the scalar slots
struct LgWqq { WqqArgs w; float* part; int ntiles, tiles_1d; };

__global__ __launch_bounds__(256) void k_lg_wqq(LgWqq a) {
    __shared__ float red[3 * 4];
    int t, tile;
    if (!task_tile(a.w.T, a.ntiles, t, tile)) return;
    const int m = a.w.tv.nq(t), ld = a.w.tv.nq_ld;
    const int m0 = (tile / a.tiles_1d) * GT, n0 = (tile % a.tiles_1d) * GT;
    float acc[3] = {0.f, 0.f, 0.f};
    if (m0 < m && n0 < m) {
        const float* sc = a.w.scal + (size_t)t * NSCAL;
        const float os = sc[S_OS], ls = sc[S_LS], il2 = 1.f / (ls * ls);
        const float* Si = a.w.Sinv + (size_t)t * ld * ld;
        const float* D2 = a.w.D2qq + (size_t)t * ld * ld;
        float* Wo = a.w.Wqq + (size_t)t * ld * ld;
        const float* ev = a.w.tv.vec_ptr(t, V_E);
        for (int e = threadIdx.x; e < GT * GT; e += 256) {
            const int i = m0 + (e >> 6), j = n0 + (e & 63);
            if (i < m && j < m) {
                const float om = 0.5f * (Si[(size_t)i * ld + j] - ev[i] * ev[j]);
                float k0, k1, k2; const float u = D2[(size_t)i * ld + j] * il2; kappa3(a.w.tv.kind, u, k0, k1, k2);
                Wo[(size_t)i * ld + j] = a.w.dirscale * om * os * k1 * il2;
                if (i == j) acc[0] += om;
                acc[1] += om * k0;
                acc[2] += om * os * k1 * u * (-2.f / ls);
            }
        }
    }
    block_sum<3, 256>(acc, red);
    if (threadIdx.x == 0) {
        float* p = a.part + ((size_t)t * a.ntiles + tile) * 4;
        p[0] = acc[0]; p[1] = acc[1]; p[2] = acc[2];
    }
}

__global__ __launch_bounds__(64) void k_lg_wqq_fin(LgWqq a) {
    const int t = blockIdx.x, lane = threadIdx.x;
    if (t >= a.w.T) return;
    float acc[3] = {0.f, 0.f, 0.f};
    float dmax = 0.f;
    for (int q = lane; q < a.ntiles; q += 64) {
        const float* p = a.part + ((size_t)t * a.ntiles + q) * 4;
        acc[0] += p[0]; acc[1] += p[1]; acc[2] += p[2];
        dmax = fmaxf(dmax, p[3]);
    }
    dmax = wave_max(dmax);
#pragma unroll
    for (int k = 0; k < 3; ++k) acc[k] = wave_sum(acc[k]);
    if (lane == 0) {
        float* sc = a.w.scal + (size_t)t * NSCAL;
        sc[S_QQ_TR] = acc[0]; sc[S_QQ_K] = acc[1]; sc[S_QQ_L] = acc[2];
    }
}

// ---- median heuristic for many points: 4 passes of an 8-bit radix select over the float bit patterns ------------------
// k_lg_med_hist: histogram of the current digit over the candidates that match the prefix found so far
//                (grid: rows are dealt to gridDim.x workgroups; LDS histogram, then integer atomics: deterministic)
// k_lg_med_pick: one thread per task walks the 256 bins, extends the prefix, clears the bins for the next pass
struct LgMedian { const float* D2ss; const int32_t* n_s; int ld; float* l0; int T; uint32_t* prefix; int* rank; int* hist; };

__device__ __forceinline__ int lg_med_shift(int pass) { return pass == 0 ? 23 : pass == 1 ? 15 : pass == 2 ? 7 : 0; }

__global__ __launch_bounds__(256) void k_lg_med_hist(LgMedian a, int pass) {
    __shared__ int h[256];
    const int t = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int n = a.n_s ? a.n_s[t] : a.ld;
    h[tid] = 0;
    __syncthreads();
    const uint32_t* D = reinterpret_cast<const uint32_t*>(a.D2ss + (size_t)t * a.ld * a.ld);
    const int shift = lg_med_shift(pass);
    const uint32_t hi_mask = pass == 0 ? 0u : ~((1u << lg_med_shift(pass - 1)) - 1u);
    const uint32_t digit_mask = pass == 3 ? 127u : 255u;
    const uint32_t prefix = pass == 0 ? 0u : a.prefix[t];
    for (int i = blockIdx.x * 4 + wv; i < n; i += gridDim.x * 4)
        for (int j = i + 1 + lane; j < n; j += 64) {
            const uint32_t v = D[(size_t)i * a.ld + j];
            if (v != 0u && (v & hi_mask) == prefix) atomicAdd(&h[(v >> shift) & digit_mask], 1);
        }
    __syncthreads();
    if (h[tid] != 0) atomicAdd(&a.hist[t * 256 + tid], h[tid]);
}

__global__ void k_lg_med_pick(LgMedian a, int pass) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.T) return;
    int* h = a.hist + t * 256;
    int rank;
    if (pass == 0) {
        int total = 0;
        for (int b = 0; b < 256; ++b) total += h[b];
        if (total == 0) { a.rank[t] = -1; a.prefix[t] = 0u; a.l0[t] = 0.f; for (int b = 0; b < 256; ++b) h[b] = 0; return; }
        rank = (total - 1) / 2;  // torch.median: lower median
    } else {
        rank = a.rank[t];
        if (rank < 0) return;
    }
    uint32_t prefix = pass == 0 ? 0u : a.prefix[t];
    int b = 0;
    for (; b < 255; ++b) {
        if (rank < h[b]) break;
        rank -= h[b];
    }
    prefix |= (uint32_t)b << lg_med_shift(pass);
    for (int q = 0; q < 256; ++q) h[q] = 0;
    a.rank[t] = rank; a.prefix[t] = prefix;
    if (pass == 3) a.l0[t] = sqrtf(0.5f * __uint_as_float(prefix));
}

}  // namespace adkf
