// Backward-stable replacement of  C = K_qs A^-1  and  alpha = A^-1 y  for ILL-CONDITIONED tasks.
//
// The pipeline keeps A^-1 explicitly (the sweep of factor.h yields it for free and the inner gradient needs all of it).
// Products with an explicit fp32 inverse carry eps32 * cond(A) * |A^-1|, and Sigma_q = K_qq - C K_sq then loses its small
// eigenvalues (~ noise) to an absolute error of ~1e-4: measured 2 % on f_out and 3e-3 on dL/dZ at cond(A) ~ 2e3 (noise
// 0.01, clustered low-dimensional features), where GPyTorch's Cholesky solves are 100x closer to float64.  Everything in
// the outer stage is a function of C and alpha (M_A = C^T Om C, M_B = -2 Om C - e alpha^T, mu = C y, Sigma_q), so solving
//       A C^T = K_sq,   A alpha = y
// by an LDL^T factorisation + two triangular substitutions for those tasks restores the factor-form accuracy (CPU
// emulation: f_out 2.6e-2 -> 2.3e-4, dL/dZ 3.5e-3 -> 5e-5).  Well-conditioned tasks - every benchmark configuration -
// leave after reading n numbers: the test is  (outputscale + noise) * max_i (A^-1)_ii > threshold,  a lower bound of cond(A).
//
// One workgroup (576 threads) per task, everything in LDS (n <= 128):
//   L   [n][n+1]   lower triangle: A, then its Schur complements column by column (right-looking, ONE barrier per pivot,
//                  columns kept unscaled until the end so that no entry is read and written in the same step)
//   R   [n][130]   up to 129 right-hand sides at a time (alpha's y rides as the first column of the first chunk); four or more
//                  lanes of one wave share a column: LDS executes a wave's accesses in order, no barrier inside the substitutions
//                  (which go four pivots per pass over the column - they are LDS-throughput-bound - and keep up to 16 loads in flight)
#pragma once
#include "problems.h"

namespace adkf {

constexpr int LDL_NT = 576;        // eight waves share columns 0..127 of a chunk, the ninth takes column 128 by itself
constexpr int LDL_RC = 129;        // right-hand sides per chunk: y + 128 query points go through in ONE pass
constexpr int LDL_RLD = LDL_RC + 1;
constexpr float LDL_THRESHOLD = 25.f;

struct LdlArgs {
    TaskView tv;
    const float* D2ss;   // [T, ns_ld, ns_ld]
    const float* D2qs;   // [T, nq_ld, ns_ld]
    const float* Ainv;   // [T, ns_ld, ns_ld]
    const float* y_s;    // [T, ns_ld]
    float* C;            // [T, nq_ld, ns_ld]   overwritten for flagged tasks
    float* vecs;         // [T, NVEC, vld]      V_ALPHA overwritten for flagged tasks
    float thresh;
    int T;
};

inline size_t ldl_smem_bytes(int ns_ld) {
    return sizeof(float) * ((size_t)ns_ld * (ns_ld + 1) + (size_t)ns_ld * LDL_RLD + ns_ld + 16);
}

__global__ __launch_bounds__(LDL_NT) void k_ldl_c(LdlArgs a) {
    extern __shared__ float ldl_sm[];
    const int t = blockIdx.x, tid = threadIdx.x;
    if (t >= a.T) return;
    const int n = a.tv.ns(t), m = a.tv.nq(t), ld = a.tv.ns_ld, LD = ld + 1;
    if (n <= 0 || m <= 0) return;
    float* L = ldl_sm;
    float* R = ldl_sm + (size_t)ld * LD;
    float* ipiv = ldl_sm + (size_t)ld * LD + (size_t)ld * LDL_RLD;
    float* red = ipiv + ld;
    const float* sc = a.tv.scal + (size_t)t * NSCAL;
    const float os = sc[S_OS], ls = sc[S_LS], noise = sc[S_NOISE], il2 = 1.f / (ls * ls);
    const int kind = a.tv.kind;

    // ---- is this task ill-conditioned?  (os + noise) max_i (A^-1)_ii <= cond(A)
    {
        const float* Ai = a.Ainv + (size_t)t * ld * ld;
        float mx = 0.f;
        for (int i = tid; i < n; i += LDL_NT) mx = fmaxf(mx, Ai[(size_t)i * ld + i]);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o, 64));
        if ((tid & 63) == 0) red[tid >> 6] = mx;
        __syncthreads();
        mx = red[0];
        for (int w = 1; w < LDL_NT / 64; ++w) mx = fmaxf(mx, red[w]);
        if (!((os + noise) * mx > a.thresh)) return;   // uniform over the workgroup
    }

    // ---- A (lower triangle) from the squared distances
    const float* D2 = a.D2ss + (size_t)t * ld * ld;
    for (int e = tid; e < n * n; e += LDL_NT) {
        const int i = e / n, j = e - i * n;
        if (j <= i) L[i * LD + j] = os * kappa0(kind, D2[(size_t)i * ld + j] * il2) + (i == j ? noise : 0.f);
    }
    __syncthreads();

    // ---- right-looking LDL^T: after step k column k holds the (unscaled) Schur-complement column c_ik, L[k][k] = pivot p_k
    {
        const int tr = tid >> 5, tc = tid & 31;
        for (int k = 0; k < n - 1; ++k) {
            const float ip = 1.f / L[k * LD + k];
            for (int i = k + 1 + tr; i < n; i += LDL_NT / 32) {
                const float f = L[i * LD + k] * ip;
                int j = k + 1 + tc;
                for (; j + 96 <= i; j += 128) {           // four independent entries in flight
                    float c[4], v[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { c[u] = L[(j + 32 * u) * LD + k]; v[u] = L[i * LD + j + 32 * u]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) L[i * LD + j + 32 * u] = fmaf(-f, c[u], v[u]);
                }
                for (; j <= i; j += 32) L[i * LD + j] -= f * L[j * LD + k];
            }
            __syncthreads();
        }
    }
    // unit-lower factor: l_ik = c_ik / p_k;  ipiv[k] = 1 / p_k
    for (int e = tid; e < n * n; e += LDL_NT) {
        const int i = e / n, k = e - i * n;
        if (k < i) L[i * LD + k] = L[i * LD + k] / L[k * LD + k];
    }
    __syncthreads();
    if (tid < n) ipiv[tid] = 1.f / L[tid * LD + tid];
    __syncthreads();

    // ---- right-hand sides in chunks: column 0 of chunk 0 is y (-> alpha), the others are columns of K_sq (-> rows of C)
    const float* Dqs = a.D2qs + (size_t)t * a.tv.nq_ld * ld;
    const float* y = a.y_s + (size_t)t * ld;
    float* Co = a.C + (size_t)t * a.tv.nq_ld * ld;
    float* alpha = a.vecs + ((size_t)t * NVEC + V_ALPHA) * a.tv.vld;
    const int total = m + 1;                      // logical right-hand sides: 0 = y, 1 + j = query j
    for (int c0 = 0; c0 < total; c0 += LDL_RC) {
        const int cols = min(LDL_RC, total - c0);
        for (int e = tid; e < cols * n; e += LDL_NT) {
            const int cc = e / n, k = e - cc * n, g = c0 + cc;
            R[k * LDL_RLD + cc] = (g == 0) ? y[k] : os * kappa0(kind, Dqs[(size_t)(g - 1) * ld + k] * il2);
        }
        __syncthreads();
        // lanes per column: 4 for a full chunk, up to a whole wave when few columns are left (the substitutions are n
        // dependent steps whatever the width, so a narrow tail must not run on four lanes); the ninth wave serves column 128
        int lpc, cc, sub;
        if (tid >= 512) {
            lpc = 64; cc = 128; sub = tid - 512;
        } else {
            const int cm = min(cols, 128);
            lpc = 4;
            while (lpc < 64 && cm * lpc * 2 <= 512) lpc *= 2;
            cc = tid / lpc; sub = tid % lpc;
            if (cc >= cm) cc = LDL_RC;        // idle
        }
        float* Rp = R;
        if (cc < cols) {
            // forward: z = L^-1 b, FOUR pivots per pass over the column (the substitutions are LDS-throughput-bound: per row
            // and 4 pivots this reads R once, L four times and writes R once instead of 12 accesses).  Lanes of one column sit
            // in one wave, which executes its LDS accesses in order: the compiler barriers keep each pass behind the stores of
            // the previous one, nothing else is needed.
            float* Rc = Rp + cc;
            int k = 0;
            for (; k + 4 <= n; k += 4) {
                asm volatile("" ::: "memory");
                const float* L1 = L + (k + 1) * LD + k; const float* L2 = L + (k + 2) * LD + k; const float* L3 = L + (k + 3) * LD + k;
                const float z0 = Rc[k * LDL_RLD];
                const float z1 = Rc[(k + 1) * LDL_RLD] - L1[0] * z0;
                const float z2 = Rc[(k + 2) * LDL_RLD] - L2[0] * z0 - L2[1] * z1;
                const float z3 = Rc[(k + 3) * LDL_RLD] - L3[0] * z0 - L3[1] * z1 - L3[2] * z2;
                asm volatile("" ::: "memory");
                if (sub == 0) { Rc[(k + 1) * LDL_RLD] = z1; Rc[(k + 2) * LDL_RLD] = z2; Rc[(k + 3) * LDL_RLD] = z3; }
                int i = k + 4 + sub;
                for (; i + 3 * lpc < n; i += 4 * lpc) {   // four independent rows in flight
                    float r[4], l[4][4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        r[u] = Rc[(i + u * lpc) * LDL_RLD];
#pragma unroll
                        for (int q = 0; q < 4; ++q) l[u][q] = L[(i + u * lpc) * LD + k + q];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u)
                        Rc[(i + u * lpc) * LDL_RLD] = r[u] - (l[u][0] * z0 + l[u][1] * z1) - (l[u][2] * z2 + l[u][3] * z3);
                }
                for (; i < n; i += lpc) {
                    const float* Li = L + i * LD + k;
                    Rc[i * LDL_RLD] -= (Li[0] * z0 + Li[1] * z1) + (Li[2] * z2 + Li[3] * z3);
                }
            }
            for (; k < n - 1; ++k) {                      // the last n mod 4 pivots
                asm volatile("" ::: "memory");
                const float zk = Rc[k * LDL_RLD];
                for (int i = k + 1 + sub; i < n; i += lpc) Rc[i * LDL_RLD] -= L[i * LD + k] * zk;
            }
            asm volatile("" ::: "memory");
            // D^-1
            for (int q = sub; q < n; q += lpc) Rc[q * LDL_RLD] *= ipiv[q];
            // backward: x = L^-T z.  Blocks of four are aligned to the top; the n mod 4 bottom... top rows go one by one first.
            const int nb = n & ~3;                        // rows [0, nb) in blocks of 4, rows [nb, n) singly
            for (int q = n - 2; q >= nb && q >= 0; --q) {
                asm volatile("" ::: "memory");
                float sum = 0.f;
                for (int i = q + 1 + sub; i < n; i += lpc) sum += L[i * LD + q] * Rc[i * LDL_RLD];
                for (int o = lpc >> 1; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
                if (sub == 0) Rc[q * LDL_RLD] -= sum;
            }
            for (int kb = nb - 4; kb >= 0; kb -= 4) {
                asm volatile("" ::: "memory");
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
                int i = kb + 4 + sub;
                for (; i + 3 * lpc < n; i += 4 * lpc) {
                    float r[4], l[4][4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        r[u] = Rc[(i + u * lpc) * LDL_RLD];
#pragma unroll
                        for (int q = 0; q < 4; ++q) l[u][q] = L[(i + u * lpc) * LD + kb + q];
                    }
#pragma unroll
                    for (int u = 0; u < 4; ++u) { s0 = fmaf(l[u][0], r[u], s0); s1 = fmaf(l[u][1], r[u], s1); s2 = fmaf(l[u][2], r[u], s2); s3 = fmaf(l[u][3], r[u], s3); }
                }
                for (; i < n; i += lpc) {
                    const float* Li = L + i * LD + kb; const float r = Rc[i * LDL_RLD];
                    s0 = fmaf(Li[0], r, s0); s1 = fmaf(Li[1], r, s1); s2 = fmaf(Li[2], r, s2); s3 = fmaf(Li[3], r, s3);
                }
                for (int o = lpc >> 1; o > 0; o >>= 1) {
                    s0 += __shfl_xor(s0, o, 64); s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); s3 += __shfl_xor(s3, o, 64);
                }
                const float* L1 = L + (kb + 1) * LD + kb; const float* L2 = L + (kb + 2) * LD + kb; const float* L3 = L + (kb + 3) * LD + kb;
                const float x3 = Rc[(kb + 3) * LDL_RLD] - s3;
                const float x2 = Rc[(kb + 2) * LDL_RLD] - s2 - L3[2] * x3;
                const float x1 = Rc[(kb + 1) * LDL_RLD] - s1 - L2[1] * x2 - L3[1] * x3;
                const float x0 = Rc[kb * LDL_RLD] - s0 - L1[0] * x1 - L2[0] * x2 - L3[0] * x3;
                asm volatile("" ::: "memory");
                if (sub == 0) { Rc[kb * LDL_RLD] = x0; Rc[(kb + 1) * LDL_RLD] = x1; Rc[(kb + 2) * LDL_RLD] = x2; Rc[(kb + 3) * LDL_RLD] = x3; }
            }
        }
        __syncthreads();
        for (int e = tid; e < cols * n; e += LDL_NT) {
            const int c2 = e / n, k = e - c2 * n, g = c0 + c2;
            const float x = R[k * LDL_RLD + c2];
            if (g == 0) alpha[k] = x; else Co[(size_t)(g - 1) * ld + k] = x;
        }
        __syncthreads();
    }
}

}  // namespace adkf
