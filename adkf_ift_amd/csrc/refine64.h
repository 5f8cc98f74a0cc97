// k_refine64: the float64 path of ILL-CONDITIONED tasks.
//
// The pipeline keeps A^-1 and Sigma_q^-1 as explicit float32 matrices (the sweep yields them for free and the closed
// forms need every entry).  That is accurate to eps32 * cond: fine for every benchmark configuration (noise 0.1, or features
// of dimension >= 64: pivot ratios below 10), wrong by 1e-3 .. 1e-2 on dL/dZ, grad_phi f_out and v for regression tasks
// with noise ~0.01 on clustered low-dimensional features (cond 1e3 .. 5e3), where the reference - GPyTorch: Cholesky +
// triangular solves, fs_mol/models/adaptive_dkt.py:183-191 - stays at 1e-5 .. 1e-4.  A CPU emulation of the precision
// choices (tools/history/emulate_precision.py, 180 random tasks) shows what it takes to reach 1e-4 everywhere: A^-1, alpha, the
// Hessian traces, C = K_qs A^-1, Sigma_q, Sigma_q^-1 and e COMPUTED in float64 and only then rounded to float32 for the
// cotangent algebra (an explicit inverse ROUNDED to float32 is accurate entry by entry, which is what the element-wise
// products with kappa' need; the float32 arithmetic that produced it was the problem).  Mixed-precision refinement of
// the float32 inverses alone is not enough, nor is a float32 LDL^T re-solve of C and alpha (round 1's ldl.h, replaced by
// this file).
//
// So: after the float32 stages have run for everybody, ONE workgroup per flagged task redoes, in float64 from the
// float32 FEATURES (squared distances in the difference form: r64_distances), everything up to the cotangent stage
//     A -> Cholesky -> A^-1, alpha, log|A| -> P = A^-1 G, beta, gamma, delta, the nine traces -> 3 x 3 Hessian
//     C = K_qs A^-1, mu, r, S = K_qq - C K_sq + noise I -> Cholesky -> S^-1, e, log|S|, f_out, C^T e
// and overwrites the float32 buffers the later kernels read; k_cotangent64 (end of this file) then does the same for the
// cotangent stage and dL/dZ.  A task is flagged when the pivot ratio max d_k / min d_k of
// its sweep (a lower bound of the condition number; <= (s + noise) / noise) exceeds `thresh` for A or for Sigma_q; tasks on
// the blocked path (> 128 points, no pivot ratio at hand) are flagged by (s + noise) / noise itself.  Everything lives in
// a float64 region of the caller's workspace (L2-resident); plain loops - this is the slow path, it only has to be right.
#pragma once
#include <utility>

#include "kernels.h"

namespace adkf {

constexpr int R64_NT = 512;
constexpr int R64_LDS_POINTS = 128;   // r64_inverse works in LDS up to this many points (n^2 doubles of dynamic shared memory)
constexpr int R64_MAXN = 1024;         // float64 region is carved for batches up to this many points (round 3: was 256; beyond R64_LDS_POINTS the
                                       // inverses run blocked, r64_inverse_blocked)
constexpr float R64_THRESHOLD = 30.f;

struct Refine64Args {
    TaskView tv;
    const float *Zs, *Zq; int d;       // the features: a flagged task takes its squared distances from them in float64 (see r64_distances)
    const float *y_s, *y_q, *priors;
    float *Ainv, *P, *C, *S;           // float32 buffers to overwrite (P, C, S may be null)
    float* vecs; float* scal; float* f_out; int32_t* info;
    float *f_in, *g_in, *gnorm;        // optional: the refined inner value / raw gradient / max |gradient| (adkf_fit, adkf_mll_value_grad)
    double* w64; size_t w64_stride;    // [T][stride] doubles
    float thresh; int T, want_hess, want_outer, lds_inverse, stop;   // lds_inverse: the launch carries R64_LDS_POINTS^2 doubles of dynamic LDS (the in-LDS inverse and the staged B operands of the products); want_outer: 0 = inner quantities only, 1 = + C and mu (prediction), 2 = + S, S^-1, e, f_out
};

// doubles per task: [A1 A2 A3 | B1 B2 | S1 S2 | 8 vectors | DDss DDqs DDqq | spare]
inline __host__ __device__ size_t r64_dd_offset(int ns, int nq) {
    return 3 * (size_t)ns * ns + 2 * (size_t)nq * ns + 2 * (size_t)nq * nq + 8 * (size_t)(ns > nq ? ns : nq);
}
// scratch of the blocked inverse behind the distances: the diagonal block, D^-1 M[K, :] and the new column panel
inline __host__ __device__ size_t r64_scratch_offset(int ns, int nq) {
    return r64_dd_offset(ns, nq) + (size_t)ns * ns + (size_t)nq * ns + (size_t)nq * nq;
}
inline size_t refine64_doubles(int ns, int nq) {
    const size_t vmax = (size_t)(ns > nq ? ns : nq);
    return r64_scratch_offset(ns, nq) + (size_t)R64_LDS_POINTS * R64_LDS_POINTS + 2 * (size_t)R64_LDS_POINTS * vmax + 64;
}

// C(i, j) = sum_k fa(i, k) fb(k, j), delivered element by element to fe(i, j, value): the O(n^3) products of the float64 path on
// the FP64 matrix pipe (v_mfma_f64_16x16x4_f64: lane l carries A[l & 15][k = l >> 4] and B[k = l >> 4][l & 15]; its C/D map is
// col = l & 15, row = (l >> 4) + 4 reg - NOT the float32 one).  The 16 x 16 output tiles are dealt round-robin to the eight
// waves; operands come straight from L2 through the caller's accessors (out-of-range rows / columns / k read as zero).
// Round 2 ran these products as one thread per output element with a scalar k loop over global memory: ~0.5 ms per 128^3
// product and workgroup, ~4 ms per launch of the path (profiles/r03_bench_*_d4.json, "before").  Barrier at the end.
typedef double f64x4_t __attribute__((ext_vector_type(4)));
template <class FA, class FB, class FE>
__device__ __forceinline__ void r64_mm_direct(int M, int N, int K, FA fa, FB fb, FE fe) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int tn = (N + 15) >> 4, tiles = ((M + 15) >> 4) * tn;
    const int lr = lane & 15, lk = lane >> 4;
    for (int tile = wv; tile < tiles; tile += R64_NT / 64) {
        const int i0 = (tile / tn) << 4, j0 = (tile % tn) << 4;
        const int ar = i0 + lr, bc = j0 + lr;
        const bool aok = ar < M, bok = bc < N;
        f64x4_t acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
        for (int k0 = 0; k0 < K; k0 += 4) {
            const int k = k0 + lk;
            const double av = (aok && k < K) ? fa(ar, k) : 0.0;
            const double bv = (bok && k < K) ? fb(k, bc) : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = i0 + lk + 4 * r, j = j0 + lr;
            if (i < M && j < N) fe(i, j, acc[r]);
        }
    }
    __syncthreads();
}

// The same product with the B operand STAGED in LDS (round 4; `stage`: 128 x 128 doubles, the region the in-LDS inverse uses): per
// 128 x 128 block of B the workgroup copies it once (32 values per lane), then wave w forms rows 16 w .. 16 w + 15 of that column
// block as eight accumulator tiles - ONE operand load from L2 per lane and k step (A: the wave's own rows, shared by its eight
// tiles) instead of two per tile, eight LDS reads.  The direct version above spent ~100 us per 128^3 product waiting for L2 (two
// dependent-latency loads in front of every MFMA, every B value fetched by eight waves); this one is bounded by the FP64 matrix
// pipe (64 cycles per MFMA and SIMD: 14 us per 128^3 product) and the A loads.
template <class FA, class FB, class FE>
__device__ __forceinline__ void r64_mm_staged(int M, int N, int K, FA fa, FB fb, FE fe, double* stage) {
    constexpr int B = R64_LDS_POINTS;   // 128: the launch carries B^2 doubles of dynamic LDS whatever the batch size
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, lr = lane & 15, lk = lane >> 4;
    for (int n0 = 0; n0 < N; n0 += B)
        for (int m0 = 0; m0 < M; m0 += B) {
            const int ar = m0 + 16 * wv + lr;
            const bool aok = ar < M;
            // only the 16-column tiles and the k rows that exist are staged and multiplied: the products with a handful of columns
            // (dL/dZ with d = 4 features) or of k steps (the distances with d = 4) used to pay for all 128 x 128
            const int nt = min(8, (N - n0 + 15) >> 4), ncol = 16 * nt;
            f64x4_t acc[8];
#pragma unroll
            for (int x = 0; x < 8; ++x) acc[x] = (f64x4_t){0.0, 0.0, 0.0, 0.0};
            for (int kc = 0; kc < K; kc += B) {
                const int kend = min(B, K - kc), krows = (kend + 3) & ~3;
                __syncthreads();                                   // the previous block of B has been read by everybody
                for (int e = threadIdx.x; e < krows * ncol; e += R64_NT) {
                    const int kk = e / ncol, j = e - kk * ncol;
                    stage[kk * B + j] = (kc + kk < K && n0 + j < N) ? fb(kc + kk, n0 + j) : 0.0;
                }
                __syncthreads();
                if (16 * wv < M - m0) {                            // (wave-uniform: this wave's rows exist)
                    double av = (aok && kc + lk < K) ? fa(ar, kc + lk) : 0.0;
                    for (int k0 = 0; k0 < kend; k0 += 4) {
                        const int kn = kc + k0 + 4 + lk;
                        const double an = (aok && k0 + 4 < kend && kn < K) ? fa(ar, kn) : 0.0;   // next step's A: in flight under the MFMAs
                        const double* bp = stage + (k0 + lk) * B + lr;
#pragma unroll
                        for (int x = 0; x < 8; ++x)
                            if (x < nt) acc[x] = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bp[16 * x], acc[x], 0, 0, 0);
                        av = an;
                    }
                }
            }
            if (16 * wv < M - m0) {
#pragma unroll
                for (int x = 0; x < 8; ++x)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = m0 + 16 * wv + lk + 4 * r, j = n0 + 16 * x + lr;
                        if (x < nt && i < M && j < N) fe(i, j, acc[x][r]);
                    }
            }
        }
    __syncthreads();
}

template <class FA, class FB, class FE>
__device__ __forceinline__ void r64_mm(int M, int N, int K, FA fa, FB fb, FE fe, double* stage = nullptr) {
    if (stage) r64_mm_staged(M, N, K, fa, fb, fe, stage);
    else r64_mm_direct(M, N, K, fa, fb, fe);
}

// y_i = sum_k fa(i, k) fx(k), delivered to fe(i, value): a wave per row, the lanes along k (coalesced for a row-major matrix;
// round 2 ran the row-wise mat-vecs as a thread per row - 128 different cache lines per k step), wave-reduced.  Barrier at the end.
template <class FA, class FX, class FE>
__device__ __forceinline__ void r64_mv(int M, int K, FA fa, FX fx, FE fe) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    constexpr int NW = R64_NT / 64, RU = 4;   // four rows of a wave in flight: their loads and the six exchange steps of their reductions overlap
    for (int i0 = wv; i0 < M; i0 += NW * RU) {
        double s[RU];
#pragma unroll
        for (int u = 0; u < RU; ++u) {
            const int i = i0 + u * NW;
            s[u] = 0.0;
            if (i < M)
                for (int k = lane; k < K; k += 64) s[u] += fa(i, k) * fx(k);
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1)
#pragma unroll
            for (int u = 0; u < RU; ++u) s[u] += __shfl_xor(s[u], o, 64);
#pragma unroll
        for (int u = 0; u < RU; ++u)
            if (lane == 0 && i0 + u * NW < M) fe(i0 + u * NW, s[u]);
    }
    __syncthreads();
}

// y_i = sum_k ft(k, i) fx(k) with the threads along i (coalesced for the COLUMNS of a row-major matrix, i.e. for M^T x - and for M x when
// M is symmetric: A^-1, G, S^-1, W_ss, W_qq): no cross-lane reduction at all - the four quarters of the workgroup take k = q, q + 4, ...
// of 128 outputs at a time and their partial sums are added in a fixed order through `part` (512 doubles of LDS).  The wave-per-row
// version above pays six exchange steps per row; seven of the path's ten mat-vecs and its three column sums run through this one.
template <class FT, class FX, class FE>
__device__ __forceinline__ void r64_mv_cols(int N, int K, FT ft, FX fx, FE fe, double* part) {
    const int il = threadIdx.x & 127, q = threadIdx.x >> 7;
    for (int i0 = 0; i0 < N; i0 += 128) {
        const int i = i0 + il;
        double s = 0.0;
        if (i < N) {
#pragma unroll 4
            for (int k = q; k < K; k += 4) s += ft(k, i) * fx(k);
        }
        __syncthreads();
        part[q * 128 + il] = s;
        __syncthreads();
        if (q == 0 && i < N) fe(i, (part[il] + part[128 + il]) + (part[256 + il] + part[384 + il]));
    }
    __syncthreads();
}

// Squared distances of a flagged task in float64, straight from the float32 features.  The GEMM form of the float32 stage
// (|x|^2 + |y|^2 - 2 x.y) carries eps32 |x|^2 into every entry: nothing for the benchmark shapes, but for clustered
// low-dimensional features (the flagged tasks) it was the LAST float32 input of the float64 path and the whole remaining error
// of it (tools/history/diag_stress.py: dL/dZ_s 1.9e-4 -> 1.7e-6 on the one stress task that stayed above 1e-4).  In float64 the same
// form carries eps64 |x|^2 - nothing - and runs on the matrix pipe (r64_mm); `same`: X == Y, the diagonal is exactly zero.
// nx / ny: scratch for the squared row norms (nx + ny doubles).
__device__ void r64_distances(const float* X, const float* Y, int nx, int ny, int d, double* out, int ldo, double* sx, double* sy, bool same, double* stage = nullptr) {
    for (int i = threadIdx.x; i < nx + ny; i += R64_NT) {
        const float* r = i < nx ? X + (size_t)i * d : Y + (size_t)(i - nx) * d;
        double s = 0.0;
        for (int c = 0; c < d; ++c) s += (double)r[c] * (double)r[c];
        if (i < nx) sx[i] = s; else sy[i - nx] = s;
    }
    __syncthreads();
    r64_mm(nx, ny, d, [=](int i, int k) { return (double)X[(size_t)i * d + k]; }, [=](int k, int j) { return (double)Y[(size_t)j * d + k]; },
       [=](int i, int j, double v) { const double q = sx[i] + sy[j] - 2.0 * v; out[(size_t)i * ldo + j] = (same && i == j) ? 0.0 : (q > 0.0 ? q : 0.0); }, stage);
}

// The exponential factor of the kernel function alone, and kappa, kappa', kappa'' from it.  A float64 exp is ~40 dependent instructions
// and the path evaluated it nine times per matrix element; up to 128 points the factors of K_ss, K_qs, K_qq are computed ONCE per launch
// and parked in the (then unused) scratch region of the blocked inverse (r64_scratch_offset: exactly ns^2 + nq ns + nq^2 doubles).
__device__ __forceinline__ double exp_term_d(int kind, double u) { return kind == 0 ? exp(-0.5 * u) : exp(-2.23606797749979 * sqrt(u)); }
__device__ __forceinline__ void kappa3_e(int kind, double u, double e, double& k0, double& k1, double& k2) {
    if (kind == 0) { k0 = e; k1 = -0.5 * e; k2 = 0.25 * e; }
    else {
        const double r = sqrt(u);
        k0 = (1.0 + 2.23606797749979 * r + (5.0 / 3.0) * u) * e;
        k1 = -(5.0 / 6.0) * (1.0 + 2.23606797749979 * r) * e;
        k2 = (25.0 / 12.0) * e;
    }
}
__device__ __forceinline__ double exp_cached(const double* E, size_t idx, int kind, double u) { return E ? E[idx] : exp_term_d(kind, u); }

__device__ __forceinline__ void kappa3_d(int kind, double u, double& k0, double& k1, double& k2) {
    if (kind == 0) { k0 = exp(-0.5 * u); k1 = -0.5 * k0; k2 = 0.25 * k0; }
    else {
        const double r = sqrt(u), e = exp(-2.23606797749979 * r);
        k0 = (1.0 + 2.23606797749979 * r + (5.0 / 3.0) * u) * e;
        k1 = -(5.0 / 6.0) * (1.0 + 2.23606797749979 * r) * e;
        k2 = (25.0 / 12.0) * e;
    }
}

// sum over the workgroup; every thread gets the total (two barriers)
__device__ __forceinline__ double r64_sum(double v, double* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    double s = 0.0;
    for (int i = 0; i < R64_NT / 64; ++i) s += red[i];
    return s;
}

// the same for NV values at once: their exchange steps overlap and the whole batch costs two barriers (red: NV * R64_NT / 64 doubles)
template <int NV>
__device__ __forceinline__ void r64_sum_n(double (&v)[NV], double* red) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)
#pragma unroll
        for (int q = 0; q < NV; ++q) v[q] += __shfl_xor(v[q], o, 64);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < NV; ++q) red[w * NV + q] = v[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        double t = 0.0;
        for (int i = 0; i < R64_NT / 64; ++i) t += red[i * NV + q];
        v[q] = t;
    }
}

// M -> M^-1 in place (n x n SPD, leading dimension ld) by Gauss-Jordan steps without pivoting, the whole workgroup; returns the
// first non-positive pivot (1-based) or 0, log|M| through logdet (the pivots are those of the LDL^T factorisation).  Every step
// is one fully parallel rank-1 update of the n^2 entries from the pivot column and the scaled pivot row, which are parked in
// LDS first: 2 n barriers in all.  (Round 2 factored M = L L^T, inverted L one column per thread and formed L^-T L^-1: with
// 128 points that left 384 of the 512 threads idle through a dependent O(n^2) chain of global loads per column - the float64
// path cost ~4 ms per launch, i.e. ONE flagged task multiplied the step time by nine; profiles/r03_bench_*_d4.json.)
// Up to 128 points the matrix makes its n steps in REGISTERS (round 4, second session): thread (ri = tid / 16, cj = tid % 16) holds the
// 4 x 8 block of rows 4 ri .. 4 ri + 3 and columns 8 cj .. 8 cj + 7; per step the owners publish pivot row and pivot column to LDS (two
// buffers in turn: one barrier per step), everybody reads its 8 row entries and its 4 column entries and makes 32 FMAs.  The in-LDS
// version below ran every step as 32 dependent read-modify-writes per thread with per-element branches and index stepping - 4.6 us per
// step, 600 us per inverse, 57 % of the float64 path (tools/history/r64_phases.sh).  A first register version with one row strip of 32 columns
// per thread still took 1.55 us per step: every thread read 33 doubles per step, 128 KB through the CU's LDS port (tools/history/r64_inv_bench.hip);
// the 4 x 8 block reads 13.  The position of the pivot inside a thread's block must be a compile-time constant (a run-time register
// index would put the block into scratch memory), hence the steps of a chunk of 32 pivots as a template pack.
// `buf`: 640 doubles of LDS, 16-byte aligned (two row / column buffers and the pivots).
template <int KK>
__device__ __forceinline__ void r64_gj_step(double (&m)[4][8], int kc, int n, int ri, int cj, double* buf, int& bad, double* piv) {
    constexpr int AK = KK & 3, BK = KK & 7;            // the pivot's row / column inside its owners' blocks
    const int k = 32 * kc + KK;
    if (k >= n) return;                                // (uniform)
    double* cb = buf + 256 * (k & 1);
    double* rb = cb + 128;
    const bool rowk = ri == (k >> 2), colk = cj == (k >> 3);
    if (colk) {
#pragma unroll
        for (int a = 0; a < 4; ++a) cb[4 * ri + a] = m[a][BK];
    }
    if (rowk) {
#pragma unroll
        for (int b = 0; b < 8; ++b) rb[8 * cj + b] = m[AK][b];
    }
    __syncthreads();                                   // (the other buffer was last read before the previous barrier)
    // (all thirteen LDS reads of the step are issued before the reciprocal starts: its ~150 cycles of dependent operations cover their latency)
    double rv[8], cr[4];
    const double p = rb[k];
#pragma unroll
    for (int b = 0; b < 8; ++b) rv[b] = rb[8 * cj + b];
#pragma unroll
    for (int a = 0; a < 4; ++a) cr[a] = cb[4 * ri + a];
    __builtin_amdgcn_sched_barrier(0);
    if (!(p > 0.0) && !bad) bad = k + 1;
    const double ps = p > 0.0 ? p : 1.0, r = 1.0 / ps;
    if (threadIdx.x == 0) piv[k] = ps;                 // (the logarithms are taken after the last step, one thread per pivot)
#pragma unroll
    for (int a = 0; a < 4; ++a) cr[a] *= r;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        if (a == AK && rowk) {
#pragma unroll
            for (int b = 0; b < 8; ++b) m[a][b] = rv[b] * r;
        } else {
#pragma unroll
            for (int b = 0; b < 8; ++b) m[a][b] = fma(-cr[a], rv[b], m[a][b]);
        }
    }
    if (colk) {
#pragma unroll
        for (int a = 0; a < 4; ++a) m[a][BK] = (a == AK && rowk) ? r : -cr[a];
    }
}
template <int... KK>
__device__ __forceinline__ void r64_gj_chunk(std::integer_sequence<int, KK...>, double (&m)[4][8], int kc, int n, int ri, int cj, double* buf, int& bad, double* piv) {
    (r64_gj_step<KK>(m, kc, n, ri, cj, buf, bad, piv), ...);
}

__device__ int r64_inverse_reg(double* M, int n, int ld, double& logdet, double* buf) {
    const int tid = threadIdx.x, ri = tid >> 4, cj = tid & 15;
    double m[4][8];
    __syncthreads();   // the callers fill M with another thread-to-element map (the in-LDS version read back its own elements)
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) { const int i = 4 * ri + a, j = 8 * cj + b; m[a][b] = (i < n && j < n) ? M[(size_t)i * ld + j] : 0.0; }
    int bad = 0;
    double* piv = buf + 512;            // [128] the pivots
#pragma unroll 1
    for (int kc = 0; kc < 4; ++kc) {
        if (32 * kc >= n) break;
        r64_gj_chunk(std::make_integer_sequence<int, 32>{}, m, kc, n, ri, cj, buf, bad, piv);
    }
    __syncthreads();
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 8; ++b) { const int i = 4 * ri + a, j = 8 * cj + b; if (i < n && j < n) M[(size_t)i * ld + j] = m[a][b]; }
    const double ld_acc = tid < n ? log(piv[tid]) : 0.0;
    logdet = r64_sum(ld_acc, buf);      // (two barriers inside: the stores above are visible to the workgroup afterwards)
    return bad;
}

__device__ int r64_inverse(double* M, int n, int ld, double& logdet, double* colv, double* rowv, double* lds = nullptr, bool in_registers = true) {
    const int tid = threadIdx.x;
    if (in_registers && n <= R64_LDS_POINTS && rowv == colv + R64_MAXN) return r64_inverse_reg(M, n, ld, logdet, colv);   // (colv, rowv: one 2 x 1024 array)
    // up to 128 points the matrix makes its n steps in LDS (`lds`: n * n doubles of dynamic shared memory, 128 KB at n = 128):
    // in global memory every step is 32 dependent read-modify-writes per thread at L2 latency - 1.3 ms per inverse, measured
    double* W = M;
    int lw = ld;
    if (lds && n <= R64_LDS_POINTS) {
        for (int e = tid; e < n * n; e += R64_NT) { const int i = e / n, j = e - i * n; lds[e] = M[(size_t)i * ld + j]; }
        W = lds; lw = n;
    }
    int bad = 0;
    double ld_acc = 0.0;
    for (int k = 0; k < n; ++k) {
        __syncthreads();                                   // the update of step k - 1 is complete
        const double p = W[(size_t)k * lw + k];
        if (!(p > 0.0) && !bad) bad = k + 1;
        const double ps = p > 0.0 ? p : 1.0, r = 1.0 / ps;
        if (tid == (k & (R64_NT - 1))) ld_acc += log(ps);   // (ONE thread per pivot takes the logarithm - a float64 log in every thread at every step was a tenth of the kernel)
        for (int i = tid; i < n; i += R64_NT) { colv[i] = W[(size_t)i * lw + k]; rowv[i] = W[(size_t)k * lw + i] * r; }
        __syncthreads();
        // (element e = tid + 512 it is (i, j) with i, j stepped instead of divided: an integer division by the runtime n per
        // element was most of this loop's instructions)
        int i = tid / n, j = tid - i * n;
        const int di = R64_NT / n, dj = R64_NT - di * n;
#pragma unroll 4
        for (int e = tid; e < n * n; e += R64_NT) {
            double* mp = W + (size_t)i * lw + j;
            if (i == k) *mp = (j == k) ? r : rowv[j];
            else if (j == k) *mp = -colv[i] * r;
            else *mp -= colv[i] * rowv[j];
            i += di; j += dj;
            if (j >= n) { j -= n; ++i; }
        }
    }
    __syncthreads();
    if (W != M) {
        for (int e = tid; e < n * n; e += R64_NT) { const int i = e / n, j = e - i * n; M[(size_t)i * ld + j] = lds[e]; }
        __syncthreads();
    }
    logdet = r64_sum(ld_acc, colv);   // (colv is free again; r64_sum needs R64_NT / 64 doubles)
    return bad;
}

// The same for more than R64_LDS_POINTS points: block Gauss-Jordan with 128-pivot blocks - the diagonal block inverted in LDS by
// r64_inverse (lds = null, i.e. the 128 KB opt-in was refused: in global memory, like r64_inverse itself then), the panel products and the rank-128 update on the FP64 matrix pipe (r64_mm).  `scr`: NB^2 + 2 NB n doubles.
//   M[K,K] <- D^-1,   M[K,j] <- D^-1 M[K,j],   M[i,K] <- -M[i,K] D^-1,   M[i,j] <- M[i,j] - M[i,K] D^-1 M[K,j]     (i, j outside K)
__device__ int r64_inverse_blocked(double* M, int n, int ld, double& logdet, double* colv, double* rowv, double* lds, double* scr) {
    constexpr int NB = R64_LDS_POINTS;
    const int tid = threadIdx.x;
    double* Db = scr;                           // [nb][nb]
    double* F = Db + (size_t)NB * NB;           // [nb][n]
    double* Tm = F + (size_t)NB * n;            // [n][nb]
    int bad = 0;
    double ld_acc = 0.0;
    for (int k0 = 0; k0 < n; k0 += NB) {
        const int nb = n - k0 < NB ? n - k0 : NB;
        __syncthreads();
        for (int e = tid; e < nb * nb; e += R64_NT) { const int i = e / nb, j = e - i * nb; Db[e] = M[(size_t)(k0 + i) * ld + k0 + j]; }
        __syncthreads();
        double ldp;
        const int b = r64_inverse(Db, nb, nb, ldp, colv, rowv, lds);
        if (b && !bad) bad = k0 + b;
        ld_acc += ldp;
        r64_mm(nb, n, nb, [=](int i, int k) { return Db[(size_t)i * nb + k]; }, [=](int k, int j) { return M[(size_t)(k0 + k) * ld + j]; },
               [=](int i, int j, double v) { F[(size_t)i * n + j] = v; });
        r64_mm(n, nb, nb, [=](int i, int k) { return M[(size_t)i * ld + k0 + k]; }, [=](int k, int j) { return Db[(size_t)k * nb + j]; },
               [=](int i, int j, double v) { Tm[(size_t)i * nb + j] = -v; });
        r64_mm(n, n, nb, [=](int i, int k) { return M[(size_t)i * ld + k0 + k]; }, [=](int k, int j) { return F[(size_t)k * n + j]; },
               [=](int i, int j, double v) {   // (reads columns K and F only, writes outside K: no hazard between the tiles)
                   if ((i < k0 || i >= k0 + nb) && (j < k0 || j >= k0 + nb)) M[(size_t)i * ld + j] -= v;
               });
        for (int e = tid; e < nb * n; e += R64_NT) {
            const int i = e / n, j = e - i * n;
            if (j < k0 || j >= k0 + nb) M[(size_t)(k0 + i) * ld + j] = F[e];
        }
        for (int e = tid; e < n * nb; e += R64_NT) {
            const int i = e / nb, j = e - i * nb;
            M[(size_t)i * ld + k0 + j] = (i >= k0 && i < k0 + nb) ? Db[(size_t)(i - k0) * nb + j] : Tm[e];
        }
    }
    __syncthreads();
    logdet = ld_acc;
    return bad;
}

// Diagnostic (adkf_double_path_tasks): which tasks of the last adkf_ift_hypergrad / adkf_outer_nll_value_grad on this workspace took
// the float64 path - the same test as in k_refine64 (level 2) below.
__global__ void k_double_path_tasks(const float* scal, int ld, int ldq, float thresh, int T, int32_t* flagged) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    const float* sc = scal + (size_t)t * NSCAL;
    const float bound = (sc[S_OS] + sc[S_NOISE]) / sc[S_NOISE];
    const float ra = sc[S_PIVR_A], rs = sc[S_PIVR_S];   // (the blocked path writes both as well: large.h)
    flagged[t] = (ra > thresh || rs > thresh) ? 1 : 0;
}

#define R64_STOP(k_) do { if (a.stop == (k_)) return; } while (0)   // uniform over the workgroup

extern __shared__ double r64_lds[];                     // R64_LDS_POINTS^2 doubles when the batch has at most that many points, else nothing

// The body of k_refine64 for the task of this workgroup (all threads call; returns early - uniformly - for unflagged tasks).
__device__ __forceinline__ void refine64_task(const Refine64Args& a) {
    __shared__ double red[R64_NT / 64];
    __shared__ __attribute__((aligned(16))) double gj_s[2 * R64_MAXN];   // pivot column / scaled pivot row of r64_inverse (first 512: the two buffers of r64_inverse_reg)
    double* const gjc = gj_s;
    double* const gjr = gj_s + R64_MAXN;
    const int t = blockIdx.x, tid = threadIdx.x;
    if (t >= a.T) return;
    const int n = a.tv.ns(t), m = a.want_outer ? a.tv.nq(t) : 0, ld = a.tv.ns_ld, ldq = a.tv.nq_ld, vld = a.tv.vld;
    if (n <= 0) return;
    float* sc = a.scal + (size_t)t * NSCAL;
    const double noise = sc[S_NOISE], os = sc[S_OS], ls = sc[S_LS], il2 = 1.0 / (ls * ls);
    const int kind = a.tv.kind;
    // ---- flagged?
    const float bound = (float)((os + noise) / noise);
    const float ra = sc[S_PIVR_A];                 // pivot ratio of the sweep of A: the register path (inner.h) or the blocked one (large.h)
    const float rs = a.want_outer < 2 ? 0.f : sc[S_PIVR_S];
    if (!(ra > a.thresh || rs > a.thresh)) return;       // uniform over the workgroup

    double* const stage = a.lds_inverse ? r64_lds : nullptr;   // B operands of the products go through LDS (r64_mm_staged)
    double* W = a.w64 + (size_t)t * a.w64_stride;
    double* A1 = W;                                 // A -> L -> A^-1
    double* A2 = A1 + (size_t)ld * ld;              // L^-1, then G
    double* A3 = A2 + (size_t)ld * ld;              // P
    double* B1 = A3 + (size_t)ld * ld;              // K_qs
    double* B2 = B1 + (size_t)ldq * ld;             // C
    double* S1 = B2 + (size_t)ldq * ld;             // S -> L_S -> S^-1
    double* S2 = S1 + (size_t)ldq * ldq;            // L_S^-1
    const int vmax = ld > ldq ? ld : ldq;
    double* v_al = S2 + (size_t)ldq * ldq;          // alpha, beta, gamma, delta, r, e, mu, spare
    double* v_be = v_al + vmax; double* v_ga = v_be + vmax; double* v_de = v_ga + vmax;
    double* v_r = v_de + vmax; double* v_e = v_r + vmax; double* v_mu = v_e + vmax;
    double* DDss = W + r64_dd_offset(ld, ldq);      // float64 squared distances (k_cotangent64 reads them again)
    double* DDqs = DDss + (size_t)ld * ld;
    double* DDqq = DDqs + (size_t)ldq * ld;
    // parked exponential factors (see exp_term_d); null beyond 128 points, where the region is the blocked inverse's scratch
    double* const Ess = (ld <= R64_LDS_POINTS && ldq <= R64_LDS_POINTS) ? W + r64_scratch_offset(ld, ldq) : nullptr;
    double* const Eqs = Ess ? Ess + (size_t)ld * ld : nullptr;
    double* const Eqq = Ess ? Eqs + (size_t)ldq * ld : nullptr;
    const float* Zs = a.Zs + (size_t)t * ld * a.d;
    r64_distances(Zs, Zs, n, n, a.d, DDss, ld, gjc, gjr, true, stage);
    if (m > 0) {
        const float* Zq = a.Zq + (size_t)t * ldq * a.d;
        r64_distances(Zq, Zs, m, n, a.d, DDqs, ld, gjc, gjr, false, stage);
        r64_distances(Zq, Zq, m, m, a.d, DDqq, ldq, gjc, gjr, true, stage);
    }
    __syncthreads();
    R64_STOP(1);   // distances
    const float* ys = a.y_s + (size_t)t * ld;
    float* vb = a.vecs + (size_t)t * NVEC * vld;

    // ---- A, Cholesky, A^-1, alpha
    for (int e = tid; e < n * n; e += R64_NT) {
        const int i = e / n, j = e % n;
        const double u = DDss[(size_t)i * ld + j] * il2, ex = exp_term_d(kind, u);
        if (Ess) Ess[(size_t)i * ld + j] = ex;
        double k0, k1, k2; kappa3_e(kind, u, ex, k0, k1, k2);
        A1[(size_t)i * ld + j] = os * k0 + (i == j ? noise : 0.0);
    }
    double logdetA;
    double* scr = W + r64_scratch_offset(ld, ldq);
    const int badA = n <= R64_LDS_POINTS ? r64_inverse(A1, n, ld, logdetA, gjc, gjr, a.lds_inverse ? r64_lds : nullptr, a.stop != -1)
                                         : r64_inverse_blocked(A1, n, ld, logdetA, gjc, gjr, a.lds_inverse ? r64_lds : nullptr, scr);
    R64_STOP(2);   // + A, A^-1
    float* Ai32 = a.Ainv + (size_t)t * ld * ld;
    for (int e = tid; e < n * n; e += R64_NT) { const int i = e / n, j = e % n; Ai32[(size_t)i * ld + j] = (float)A1[(size_t)i * ld + j]; }
    r64_mv_cols(n, n, [=](int k, int i) { return A1[(size_t)k * ld + i]; }, [=](int k) { return (double)ys[k]; },
                [=](int i, double v) { v_al[i] = v; vb[V_ALPHA * vld + i] = (float)v; }, gjc);   // (A^-1 is symmetric)

    R64_STOP(3);   // + copy-out, alpha
    // ---- inner scalars and the 3 x 3 Hessian (oracle/closed_form.py::inner_stage)
    {
        for (int e = tid; e < n * n; e += R64_NT) {   // G = dK/dl
            const int i = e / n, j = e % n;
            const double u = DDss[(size_t)i * ld + j] * il2;
            double k0, k1, k2; kappa3_e(kind, u, exp_cached(Ess, (size_t)i * ld + j, kind, u), k0, k1, k2);
            A2[(size_t)i * ld + j] = os * k1 * u * (-2.0 / ls);
        }
        __syncthreads();
        r64_mv_cols(n, n, [=](int k, int i) { return A2[(size_t)k * ld + i]; }, [=](int k) { return v_al[k]; }, [=](int i, double v) { v_be[i] = v; }, gjc);
        r64_mv_cols(n, n, [=](int k, int i) { return A1[(size_t)k * ld + i]; }, [=](int k) { return v_al[k]; }, [=](int i, double v) { v_ga[i] = v; }, gjc);
        r64_mv_cols(n, n, [=](int k, int i) { return A1[(size_t)k * ld + i]; }, [=](int k) { return v_be[k]; }, [=](int i, double v) { v_de[i] = v; }, gjc);
        if (a.want_hess) {
            __syncthreads();
            r64_mm(n, n, n, [=](int i, int k) { return A1[(size_t)i * ld + k]; }, [=](int k, int j) { return A2[(size_t)k * ld + j]; },
                   [=](int i, int j, double v) { A3[(size_t)i * ld + j] = v; }, stage);   // P = A^-1 G
        }
        __syncthreads();
        R64_STOP(4);   // + G, three mat-vecs, P = A^-1 G
        double trAinv = 0, trAinvG = 0, aGa = 0, trA2 = 0, trPA = 0, trPP = 0, trAinvKll = 0, aKlla = 0;
        for (int e = tid; e < n * n; e += R64_NT) {
            const int i = e / n, j = e % n;
            const double ai = A1[(size_t)i * ld + j], g = A2[(size_t)i * ld + j];
            if (i == j) trAinv += ai;
            trAinvG += ai * g; aGa += v_al[i] * v_al[j] * g; trA2 += ai * ai;
            if (a.want_hess) {
                const double pij = A3[(size_t)i * ld + j], pji = A3[(size_t)j * ld + i];
                const double u = DDss[(size_t)i * ld + j] * il2;
                double k0, k1, k2; kappa3_e(kind, u, exp_cached(Ess, (size_t)i * ld + j, kind, u), k0, k1, k2);
                const double Kll = os * (k2 * 4.0 * u * u + k1 * 6.0 * u) * il2;
                trPA += pij * ai; trPP += pij * pji; trAinvKll += ai * Kll; aKlla += v_al[i] * v_al[j] * Kll;
            }
        }
        double aa = 0, ya = 0, ag = 0, bg = 0, bd = 0, ab = 0;
        for (int i = tid; i < n; i += R64_NT) {
            aa += v_al[i] * v_al[i]; ya += (double)ys[i] * v_al[i]; ag += v_al[i] * v_ga[i]; bg += v_be[i] * v_ga[i];
            bd += v_be[i] * v_de[i]; ab += v_al[i] * v_be[i];
        }
        {   // fourteen sums, one pair of barriers (gj_s is free here: 14 x 8 doubles)
            double sv[14] = {trAinv, trAinvG, aGa, trA2, trPA, trPP, trAinvKll, aKlla, aa, ya, ag, bg, bd, ab};
            r64_sum_n<14>(sv, gjc);
            trAinv = sv[0]; trAinvG = sv[1]; aGa = sv[2]; trA2 = sv[3]; trPA = sv[4]; trPP = sv[5]; trAinvKll = sv[6]; aKlla = sv[7];
            aa = sv[8]; ya = sv[9]; ag = sv[10]; bg = sv[11]; bd = sv[12]; ab = sv[13];
        }
        for (int i = tid; i < n; i += R64_NT) { vb[V_BETA * vld + i] = (float)v_be[i]; vb[V_GAMMA * vld + i] = (float)v_ga[i]; vb[V_DELTA * vld + i] = (float)v_de[i]; }
        if (a.want_hess && a.P) {
            float* P32 = a.P + (size_t)t * ld * ld;
            for (int e = tid; e < n * n; e += R64_NT) { const int i = e / n, j = e % n; P32[(size_t)i * ld + j] = (float)A3[(size_t)i * ld + j]; }
        }
        if (tid == 0) {
            const float* pri = a.priors + t * 4;
            const double fn = (double)n;
            const double d1[3] = {sc[S_D1N], sc[S_D1S], sc[S_D1L]}, d2[3] = {sc[S_D2N], sc[S_D2S], sc[S_D2L]};
            double lp = 0, dpn = 0, dpl = 0, d2pn = 0, d2pl = 0;
            if (pri[1] > 0.f) {
                const double lx = log(noise), s2 = (double)pri[1] * pri[1], z = (lx - pri[0]) / s2;
                lp += -lx - log((double)pri[1]) - 0.5 * 1.8378770664093453 - 0.5 * (lx - pri[0]) * z;
                dpn = (-1.0 - z) / noise; d2pn = (1.0 + z - 1.0 / s2) / (noise * noise);
            }
            if (pri[3] > 0.f) {
                const double lx = log(ls), s2 = (double)pri[3] * pri[3], z = (lx - pri[2]) / s2;
                lp += -lx - log((double)pri[3]) - 0.5 * 1.8378770664093453 - 0.5 * (lx - pri[2]) * z;
                dpl = (-1.0 - z) / ls; d2pl = (1.0 + z - 1.0 / s2) / (ls * ls);
            }
            const double nll = 0.5 * ya + 0.5 * logdetA + 0.5 * fn * 1.8378770664093453;
            const double gt0 = 0.5 * trAinv - 0.5 * aa - dpn;
            const double gt1 = (0.5 * (fn - noise * trAinv) - 0.5 * (ya - noise * aa)) / os;
            const double gt2 = 0.5 * trAinvG - 0.5 * aGa - dpl;
            sc[S_FIN] = (float)((nll - lp) / fn);
            sc[S_GIN0] = (float)(gt0 * d1[0] / fn); sc[S_GIN1] = (float)(gt1 * d1[1] / fn); sc[S_GIN2] = (float)(gt2 * d1[2] / fn);
            if (a.f_in) a.f_in[t] = badA ? INFINITY : sc[S_FIN];
            if (a.g_in) { a.g_in[t * 3 + 0] = sc[S_GIN0]; a.g_in[t * 3 + 1] = sc[S_GIN1]; a.g_in[t * 3 + 2] = sc[S_GIN2]; }
            if (a.gnorm) a.gnorm[t] = fmaxf(fabsf(sc[S_GIN0]), fmaxf(fabsf(sc[S_GIN1]), fabsf(sc[S_GIN2])));
            sc[S_LOGDET] = (float)logdetA; sc[S_TRAINV] = (float)trAinv; sc[S_AA] = (float)aa; sc[S_YA] = (float)ya;
            sc[S_TRAINVG] = (float)trAinvG; sc[S_AGA] = (float)aGa; sc[S_GT0] = (float)gt0; sc[S_GT1] = (float)gt1; sc[S_GT2] = (float)gt2;
            if (a.want_hess) {
                double h[3][3];
                h[0][0] = ag - 0.5 * trA2 - d2pn;
                h[0][1] = ((aa - noise * ag) - 0.5 * (trAinv - noise * trA2)) / os;
                h[0][2] = bg - 0.5 * trPA;
                h[1][1] = ((ya - 2.0 * noise * aa + noise * noise * ag) - 0.5 * (fn - 2.0 * noise * trAinv + noise * noise * trA2)) / (os * os);
                h[1][2] = ((ab - noise * bg) - 0.5 * (trAinvG - noise * trPA)) / os - (0.5 * aGa - 0.5 * trAinvG) / os;
                h[2][2] = bd - 0.5 * aKlla - 0.5 * trPP + 0.5 * trAinvKll - d2pl;
                h[1][0] = h[0][1]; h[2][0] = h[0][2]; h[2][1] = h[1][2];
                const double gt[3] = {gt0, gt1, gt2};
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) sc[S_H0 + i * 3 + j] = (float)((h[i][j] * d1[i] * d1[j] + (i == j ? gt[i] * d2[i] : 0.0)) / fn);
            }
            if (badA && a.info[t] == 0) a.info[t] = badA;
        }
    }
    __syncthreads();
    R64_STOP(5);   // + traces, reductions, Hessian
    if (m <= 0) return;

    // ---- outer: C, mu, r, S, S^-1, e, f_out, C^T e   (oracle/closed_form.py::outer_stage)
    // (the labels of the query points exist at level 2 only: adkf_predict's batch may carry y_q = NULL, and level 1 - C and mu -
    // does not read them)
    const float* yq = a.y_q ? a.y_q + (size_t)t * ldq : nullptr;
    for (int e = tid; e < m * n; e += R64_NT) {
        const int i = e / n, j = e % n;
        const double u = DDqs[(size_t)i * ld + j] * il2, ex = exp_term_d(kind, u);
        if (Eqs) Eqs[(size_t)i * ld + j] = ex;
        double k0, k1, k2; kappa3_e(kind, u, ex, k0, k1, k2);
        B1[(size_t)i * ld + j] = os * k0;
    }
    __syncthreads();
    r64_mm(m, n, n, [=](int i, int k) { return B1[(size_t)i * ld + k]; }, [=](int k, int j) { return A1[(size_t)k * ld + j]; },
           [=](int i, int j, double v) { B2[(size_t)i * ld + j] = v; }, stage);           // C = K_qs A^-1
    r64_mv(m, n, [=](int i, int k) { return B1[(size_t)i * ld + k]; }, [=](int k) { return v_al[k]; },
           [=](int i, double v) { v_mu[i] = v; v_r[i] = yq ? (double)yq[i] - v : 0.0; });
    if (a.C) {
        float* C32 = a.C + (size_t)t * ldq * ld;
        for (int e = tid; e < m * n; e += R64_NT) { const int i = e / n, j = e % n; C32[(size_t)i * ld + j] = (float)B2[(size_t)i * ld + j]; }
    }
    if (a.want_outer < 2) {
        for (int i = tid; i < m; i += R64_NT) vb[V_MU * vld + i] = (float)v_mu[i];
        return;
    }
    __syncthreads();
    R64_STOP(6);   // + K_qs, C, mu
    r64_mm(m, m, n, [=](int i, int k) { return B2[(size_t)i * ld + k]; }, [=](int k, int j) { return B1[(size_t)j * ld + k]; },
           [=](int i, int j, double v) {                                            // Sigma_q = K_qq + noise I - C K_sq, the lower triangle mirrored
               if (j > i) return;
               const double u = DDqq[(size_t)i * ldq + j] * il2, ex = exp_term_d(kind, u);
               if (Eqq) { Eqq[(size_t)i * ldq + j] = ex; Eqq[(size_t)j * ldq + i] = ex; }
               double k0, k1, k2; kappa3_e(kind, u, ex, k0, k1, k2);
               const double sv = os * k0 + (i == j ? noise : 0.0) - v;
               S1[(size_t)i * ldq + j] = sv; S1[(size_t)j * ldq + i] = sv;
           }, stage);
    double logdetS;
    const int badS = m <= R64_LDS_POINTS ? r64_inverse(S1, m, ldq, logdetS, gjc, gjr, a.lds_inverse ? r64_lds : nullptr, a.stop != -1)
                                         : r64_inverse_blocked(S1, m, ldq, logdetS, gjc, gjr, a.lds_inverse ? r64_lds : nullptr, scr);
    R64_STOP(7);   // + S, S^-1
    if (a.S) {
        float* S32 = a.S + (size_t)t * ldq * ldq;
        for (int e = tid; e < m * m; e += R64_NT) { const int i = e / m, j = e % m; S32[(size_t)i * ldq + j] = (float)S1[(size_t)i * ldq + j]; }
    }
    r64_mv_cols(m, m, [=](int k, int i) { return S1[(size_t)k * ldq + i]; }, [=](int k) { return v_r[k]; }, [=](int i, double v) { v_e[i] = v; }, gjc);
    double q = 0.0;
    for (int i = tid; i < m; i += R64_NT) {
        q += v_r[i] * v_e[i];
        vb[V_MU * vld + i] = (float)v_mu[i]; vb[V_R * vld + i] = (float)v_r[i]; vb[V_E * vld + i] = (float)v_e[i];
    }
    q = r64_sum(q, red);
    r64_mv_cols(n, m, [=](int k, int j) { return B2[(size_t)k * ld + j]; }, [=](int k) { return v_e[k]; },
                [=](int j, double v) { vb[V_CTE * vld + j] = (float)v; }, gjc);                                      // C^T e
    if (tid == 0) {
        const double f = 0.5 * q + 0.5 * logdetS + 0.5 * (double)m * 1.8378770664093453;
        sc[S_FOUT] = (float)f; sc[S_LOGDETS] = (float)logdetS;
        if (a.f_out) a.f_out[t] = badS ? NAN : (float)f;
        if (badS && a.info[t] == 0) a.info[t] = 100000 + badS;
    }
}

__global__ __launch_bounds__(R64_NT) void k_refine64(Refine64Args a) { refine64_task(a); }

// k_cotangent64: the cotangent stage of the SAME flagged tasks, after the float32 kernels have written theirs.
//
// With A^-1, C, Sigma_q^-1 and e computed in float64 the remaining error of flagged tasks (cond 2e3 .. 5e3) sat in the float32
// products and reductions downstream: Omega C, C^T (Omega C), (A^-1 B_v) A^-1 and the sums behind grad_phi f_out - 1.2x .. 2.3x the
// tolerance on v and dL/dZ for 4 of 180 stress tasks, none when the emulation runs them in float64 (tools/history/emulate_precision.py,
// configuration "all64w").  This kernel redoes exactly that algebra from the float64 matrices k_refine64 left in the workspace
// region - W_ss (direct and mixed part), W_qs, W_qq, grad_phi f_out, v, w - and then dL/dZ itself, in the difference form
//     dZs_i = sum_k 4 Wss_ik (z_i - z_k) + sum_q 2 Wqs_qi (z_i - zq_q),   dZq_i = sum_k 2 Wqs_ik (zq_i - z_k) + sum_q 4 Wqq_iq (zq_i - zq_q)
// (the float32 kernels form coef_i z_i - sum_k W_ik z_k: two large terms that cancel when the features are clustered, which
// is exactly when a task is flagged - one stress task stayed at 1.9x the tolerance on dL/dZ_s until this was float64 as
// well).  It runs LAST and overwrites grad_phi f_out, v and the rows of dL/dZ of its tasks.  Formulas: ProbOC / ProbMA / k_wqq /
// solve_v_task / ProbMixed / ProbDZ (problems.h, kernels.h), i.e. oracle/closed_form.py::outer_stage, ::mixed_stage,
// ::dz_from_weights.  Everybody else leaves after the flag test.
struct Cot64Args {
    TaskView tv; const float *Zs, *Zq; float *dZs, *dZq; int d; float* vecs; float* scal;
    double* w64; size_t w64_stride; float thresh; int T, with_hessian, flags; float dirscale, corrscale; float *g_phi_out, *v_out;
    float* H_out;   // [T, 9] or null: the float64 path's Hessian (k_refine64 leaves it in the scalars) for the caller
    int lds_stage;  // the launch carries R64_LDS_POINTS^2 doubles of dynamic LDS: the products stage their B operands there
    int stop;       // diagnostics (ADKF_R64_STOP, tools/history/r64_phases.sh): leave after phase `stop` (0: run everything)
};

__device__ __forceinline__ void cotangent64_task(const Cot64Args& a) {
    __shared__ double red[R64_NT / 64];
    __shared__ double coef[4];
    const int t = blockIdx.x, tid = threadIdx.x;
    if (t >= a.T) return;
    const int n = a.tv.ns(t), m = a.tv.nq(t), ld = a.tv.ns_ld, ldq = a.tv.nq_ld, vld = a.tv.vld;
    if (n <= 0 || m <= 0) return;
    float* sc = a.scal + (size_t)t * NSCAL;
    const double noise = sc[S_NOISE], os = sc[S_OS], ls = sc[S_LS], il2 = 1.0 / (ls * ls), gl = -2.0 / ls;
    const int kind = a.tv.kind;
    const float bound = (float)((os + noise) / noise);                       // the flag test of k_refine64 (level 2)
    const float ra = sc[S_PIVR_A], rs = sc[S_PIVR_S];   // (the blocked path writes both as well: large.h)
    if (!(ra > a.thresh || rs > a.thresh)) return;

    double* const stage = a.lds_stage ? r64_lds : nullptr;
    double* W = a.w64 + (size_t)t * a.w64_stride;                            // the layout of k_refine64
    double* A1 = W;                                 // A^-1
    double* A2 = A1 + (size_t)ld * ld;              // (G: spent)        -> M_A -> W_ss
    double* A3 = A2 + (size_t)ld * ld;              // P = A^-1 G  (valid when the Hessian was asked for)
    double* B1 = A3 + (size_t)ld * ld;              // (K_qs: spent)     -> Omega C -> W_qs
    double* B2 = B1 + (size_t)ldq * ld;             // C
    double* S1 = B2 + (size_t)ldq * ld;             // S^-1
    double* S2 = S1 + (size_t)ldq * ldq;            // (L_S^-1: spent)   -> W_qq
    const int vmax = ld > ldq ? ld : ldq;
    double* v_al = S2 + (size_t)ldq * ldq;
    double* v_be = v_al + vmax; double* v_ga = v_be + vmax; double* v_de = v_ga + vmax;
    double* v_w = v_de + vmax;                      // (r: written out)  -> w
    double* v_e = v_w + vmax;
    double* v_cte = v_e + vmax;                     // (mu: written out) -> C^T e
    (void)v_be;
    const double* Dss = W + r64_dd_offset(ld, ldq);  // the float64 squared distances k_refine64 computed
    const double* Dqs = Dss + (size_t)ld * ld;
    const double* Dqq = Dqs + (size_t)ldq * ld;
    // the exponential factors refine64_task parked for this launch (null beyond 128 points)
    const double* const Ess = (ld <= R64_LDS_POINTS && ldq <= R64_LDS_POINTS) ? W + r64_scratch_offset(ld, ldq) : nullptr;
    const double* const Eqs = Ess ? Ess + (size_t)ld * ld : nullptr;
    const double* const Eqq = Ess ? Eqs + (size_t)ldq * ld : nullptr;
    float* vb = a.vecs + (size_t)t * NVEC * vld;
    const double dir = a.dirscale, corr = a.with_hessian ? (double)a.corrscale : 0.0;

    __shared__ double mvbuf[512];                   // partial sums of r64_mv_cols
    r64_mv_cols(n, m, [=](int k, int j) { return B2[(size_t)k * ld + j]; }, [=](int k) { return v_e[k]; },
                [=](int j, double v) { v_cte[j] = v; }, mvbuf);                                                      // C^T e
    r64_mm(m, n, m, [=](int i, int k) { return S1[(size_t)i * ldq + k]; }, [=](int k, int j) { return B2[(size_t)k * ld + j]; },
           [=](int i, int j, double v) { B1[(size_t)i * ld + j] = 0.5 * (v - v_e[i] * v_cte[j]); }, stage);   // Omega C = (S^-1 C - e (C^T e)^T) / 2
    R64_STOP(9);   // (8 = all of refine64_task) + C^T e, Omega C
    double oc0 = 0, oc1 = 0, ma0 = 0, ma1 = 0, ma2 = 0, qq0 = 0, qq1 = 0, qq2 = 0;
    {   // M_A = C^T (Omega C) + sym(C^T e alpha^T)   (the three reductions ride in the product's epilogue)
        double* pm0 = &ma0; double* pm1 = &ma1; double* pm2 = &ma2;
        r64_mm(n, n, m, [=](int i, int k) { return B2[(size_t)k * ld + i]; }, [=](int k, int j) { return B1[(size_t)k * ld + j]; },
               [=](int i, int j, double v) {
                   const double MA = v + 0.5 * (v_cte[i] * v_al[j] + v_al[i] * v_cte[j]);
                   A2[(size_t)i * ld + j] = MA;
                   const double u = Dss[(size_t)i * ld + j] * il2;
                   double k0, k1, k2; kappa3_e(kind, u, exp_cached(Ess, (size_t)i * ld + j, kind, u), k0, k1, k2);
                   if (i == j) *pm0 += MA;
                   *pm1 += MA * k0; *pm2 += MA * os * k1 * u * gl;
               }, stage);
    }                                               // (barrier inside) Omega C has been read by everybody: it turns into W_qs in place
    R64_STOP(10);  // + M_A
    for (int e = tid; e < m * n; e += R64_NT) {     // M_B -> W_qs
        const int i = e / n, j = e % n;
        const double MB = -2.0 * B1[(size_t)i * ld + j] - v_e[i] * v_al[j];
        const double u = Dqs[(size_t)i * ld + j] * il2;
        double k0, k1, k2; kappa3_e(kind, u, exp_cached(Eqs, (size_t)i * ld + j, kind, u), k0, k1, k2);
        B1[(size_t)i * ld + j] = dir * MB * os * k1 * il2;
        oc0 += MB * k0; oc1 += MB * os * k1 * u * gl;
    }
    for (int e = tid; e < m * m; e += R64_NT) {     // Omega -> W_qq
        const int i = e / m, j = e % m;
        const double om = 0.5 * (S1[(size_t)i * ldq + j] - v_e[i] * v_e[j]);
        const double u = Dqq[(size_t)i * ldq + j] * il2;
        double k0, k1, k2; kappa3_e(kind, u, exp_cached(Eqq, (size_t)i * ldq + j, kind, u), k0, k1, k2);
        S2[(size_t)i * ldq + j] = dir * om * os * k1 * il2;
        if (i == j) qq0 += om;
        qq1 += om * k0; qq2 += om * os * k1 * u * gl;
    }
    {   // eight sums, one pair of barriers
        __shared__ double red8[8 * (R64_NT / 64)];
        double sv[8] = {oc0, oc1, ma0, ma1, ma2, qq0, qq1, qq2};
        r64_sum_n<8>(sv, red8);
        oc0 = sv[0]; oc1 = sv[1]; ma0 = sv[2]; ma1 = sv[3]; ma2 = sv[4]; qq0 = sv[5]; qq1 = sv[6]; qq2 = sv[7];
    }
    if (tid == 0) {                                 // grad_phi f_out, v = H^-1 grad (solve_v_task)
        const double d1[3] = {sc[S_D1N], sc[S_D1S], sc[S_D1L]};
        const double g[3] = {(qq0 + ma0) * d1[0], (ma1 + oc0 + qq1) * d1[1], (ma2 + oc1 + qq2) * d1[2]};
        double vd[3] = {0.0, 0.0, 0.0};
        if (a.with_hessian && !(a.flags & 1)) {
            double Mx[3][4];
            for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) Mx[i][j] = sc[S_H0 + i * 3 + j]; Mx[i][3] = g[i]; }
            for (int c = 0; c < 3; ++c) {
                int pv = c;
                for (int r = c + 1; r < 3; ++r) if (fabs(Mx[r][c]) > fabs(Mx[pv][c])) pv = r;
                if (pv != c) for (int j = 0; j < 4; ++j) { const double tmp = Mx[c][j]; Mx[c][j] = Mx[pv][j]; Mx[pv][j] = tmp; }
                const double ip = 1.0 / Mx[c][c];
                for (int r = c + 1; r < 3; ++r) { const double f = Mx[r][c] * ip; for (int j = c; j < 4; ++j) Mx[r][j] -= f * Mx[c][j]; }
            }
            for (int c = 2; c >= 0; --c) { double s = Mx[c][3]; for (int j = c + 1; j < 3; ++j) s -= Mx[c][j] * vd[j]; vd[c] = s / Mx[c][c]; }
        }
        for (int q = 0; q < 3; ++q) { sc[S_GOUT0 + q] = (float)g[q]; sc[S_V0 + q] = (float)vd[q]; }
        if (a.g_phi_out) for (int q = 0; q < 3; ++q) a.g_phi_out[t * 3 + q] = (float)g[q];
        if (a.v_out) for (int q = 0; q < 3; ++q) a.v_out[t * 3 + q] = (float)vd[q];
        if (a.H_out && a.with_hessian) for (int q = 0; q < 9; ++q) a.H_out[t * 9 + q] = sc[S_H0 + q];
        coef[0] = vd[0] * d1[0]; coef[1] = vd[1] * d1[1] / os; coef[2] = vd[2] * d1[2];
        sc[S_CN] = (float)coef[0]; sc[S_CS] = (float)coef[1]; sc[S_CL] = (float)coef[2];
    }
    __syncthreads();
    R64_STOP(11);  // + W_qs, W_qq, reductions, v
    const double cn = coef[0], cs = coef[1], cl = coef[2];
    if (a.with_hessian) {
        for (int i = tid; i < n; i += R64_NT) {
            const double w = cn * v_ga[i] + cs * (v_al[i] - noise * v_ga[i]) + cl * v_de[i];
            v_w[i] = w;
            vb[V_W * vld + i] = (float)w;
        }
    }
    __syncthreads();
    const double fn = (double)n;
    // W_ss = direct part - mixed-partial part (ProbMA / ProbMixed epilogues); ((A^-1 B_v) A^-1)_ij with
    // A^-1 B_v = (cn - cs noise) A^-1 + cs I + cl P comes out of the matrix pipe, the rest is its epilogue
    auto wss_of = [=](int i, int j, double xa) {
        const double u = Dss[(size_t)i * ld + j] * il2;
        double k0, k1, k2; kappa3_e(kind, u, exp_cached(Ess, (size_t)i * ld + j, kind, u), k0, k1, k2);
        double wss = dir * A2[(size_t)i * ld + j] * os * k1 * il2;
        if (corr != 0.0) {
            const double dgdA = (-0.5 * xa + 0.5 * (v_w[i] * v_al[j] + v_al[i] * v_w[j])) / fn;
            const double Q = 0.5 * (A1[(size_t)i * ld + j] - v_al[i] * v_al[j]) / fn;
            const double dBv = cs * os * k1 + cl * os * gl * (k1 + u * k2);
            wss -= corr * (dgdA * os * k1 * il2 + Q * dBv * il2);
        }
        A2[(size_t)i * ld + j] = wss;               // (element (i, j) of M_A is read by this lane only)
    };
    if (corr != 0.0) {
        const double ca = cn - cs * noise;
        r64_mm(n, n, n, [=](int i, int k) { return ca * A1[(size_t)i * ld + k] + (i == k ? cs : 0.0) + cl * A3[(size_t)i * ld + k]; },
               [=](int k, int j) { return A1[(size_t)k * ld + j]; }, wss_of, stage);
    } else {
        for (int e = tid; e < n * n; e += R64_NT) wss_of(e / n, e % n, 0.0);
        __syncthreads();
    }
    R64_STOP(12);  // + W_ss with the mixed part
    // ---- dL/dZ in the difference form, from the float64 weights
    const int d = a.d;
    const float* Zs = a.Zs + (size_t)t * ld * d;
    const float* Zq = a.Zq + (size_t)t * ldq * d;
    // (float64: z_i sum_k W_ik - sum_k W_ik z_k loses cond digits of sixteen, not of seven - the difference form was what the
    // float32 kernels could not afford to skip; the W Z products run on the matrix pipe)
    double* rs_ss = v_be; double* rs_qs_col = v_ga; double* rs_qs_row = v_de; double* rs_qq = v_w;   // (all spent by now)
    r64_mv_cols(n, m, [=](int k, int i) { return B1[(size_t)k * ld + i]; }, [](int) { return 1.0; }, [=](int i, double v) { rs_qs_col[i] = v; }, mvbuf);
    r64_mv_cols(n, n, [=](int k, int i) { return A2[(size_t)k * ld + i]; }, [](int) { return 1.0; }, [=](int i, double v) { rs_ss[i] = v; }, mvbuf);   // (W_ss, W_qq: symmetric)
    r64_mv(m, n, [=](int i, int k) { return B1[(size_t)i * ld + k]; }, [](int) { return 1.0; }, [=](int i, double v) { rs_qs_row[i] = v; });
    r64_mv_cols(m, m, [=](int k, int i) { return S2[(size_t)k * ldq + i]; }, [](int) { return 1.0; }, [=](int i, double v) { rs_qq[i] = v; }, mvbuf);
    R64_STOP(13);  // + row / column sums of the weights
    if (a.dZs) {
        float* out = a.dZs + (size_t)t * ld * d;
        // dZs_i = 4 (rs_ss_i z_i - sum_k Wss_ik z_k) + 2 (cs_qs_i z_i - sum_q Wqs_qi zq_q): one product over k = [support | query]
        r64_mm(n, d, n + m, [=](int i, int k) { return k < n ? 4.0 * A2[(size_t)i * ld + k] : 2.0 * B1[(size_t)(k - n) * ld + i]; },
               [=](int k, int c) { return k < n ? (double)Zs[(size_t)k * d + c] : (double)Zq[(size_t)(k - n) * d + c]; },
               [=](int i, int c, double v) { out[(size_t)i * d + c] = (float)((4.0 * rs_ss[i] + 2.0 * rs_qs_col[i]) * (double)Zs[(size_t)i * d + c] - v); }, stage);
    }
    if (a.dZq) {
        float* out = a.dZq + (size_t)t * ldq * d;
        r64_mm(m, d, n + m, [=](int i, int k) { return k < n ? 2.0 * B1[(size_t)i * ld + k] : 4.0 * S2[(size_t)i * ldq + (k - n)]; },
               [=](int k, int c) { return k < n ? (double)Zs[(size_t)k * d + c] : (double)Zq[(size_t)(k - n) * d + c]; },
               [=](int i, int c, double v) { out[(size_t)i * d + c] = (float)((2.0 * rs_qs_row[i] + 4.0 * rs_qq[i]) * (double)Zq[(size_t)i * d + c] - v); }, stage);
    }
}

// The float64 path of the hypergradient call as ONE launch at the very end of the pipeline (round 4; it used to be k_refine64 in
// the middle - its float32 roundings fed the float32 cotangent kernels of the flagged tasks - and k_cotangent64 at the end, which
// overwrote what those kernels had produced for exactly these tasks: dead work and a second launch that every unflagged batch
// paid ~5 us for).  Everybody else leaves after the flag test, once.
__global__ __launch_bounds__(R64_NT) void k_tail64(Refine64Args ra, Cot64Args ca) {
    refine64_task(ra);
    __syncthreads();
    if (ra.stop > 0 && ra.stop <= 8) return;
    cotangent64_task(ca);
}

}  // namespace adkf
