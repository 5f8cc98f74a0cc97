// k_dz: both cotangents dL/dZ_s, dL/dZ_q of ONE task in ONE workgroup, from the three weight matrices (oracle/closed_form.py::
// dz_from_weights; the chain rule through r^2_ij = |z_i - z_j|^2 / l^2 of fs_mol/utils/gp_utils.py:26-30):
//     dZs_i = coef_s[i] Zs_i - sum_k 4 Wss[i,k] Zs_k - sum_q 2 Wqs[q,i] Zq_q,      coef_s = rowsum(4 Wss) + colsum(2 Wqs)
//     dZq_i = coef_q[i] Zq_i - sum_k 2 Wqs[i,k] Zs_k - sum_q 4 Wqq[i,q] Zq_q,      coef_q = rowsum(2 Wqs) + rowsum(4 Wqq)
// Round 3 ran them as two launches of the generic 64 x 64-tile GEMM (ProbDZ): 62 + 56 us at C2 for 8.6 GFLOP, every weight tile
// staged eight times.  Here the products run like the ones of k_hyper (hyper.h: both operands in LDS, wave w owns rows
// 16 w .. 16 w + 15 of the result as eight accumulator tiles): the A image is one of 4 Wss | 2 Wqs | 4 Wqq (row-major, ld = 144, so
// that 2 Wqs serves as [row][k] for dZq AND down its columns as (2 Wqs)^T for dZs), the B image a 128-column chunk of Zs or Zq read
// down its columns; two chunks of the result (128 x 256 per cotangent = four accumulator sets) are kept in registers while the
// images rotate underneath them, so each weight image is staged ONCE per 256 feature columns and each feature chunk at most twice.
// Full 128 + 128-point batches with d a multiple of 128 (C2: d = 256); everything else keeps ProbDZ.
//
// MEASURED AND NOT SHIPPED (round 4; build with -DADKF_VARIANT_DZ=1, tools/hyper_bench.hip times it): the eight products run at
// the matrix pipe's floor (15 - 17 k cycles each, 131 k together = 62 us), but the image hand-overs between them (barrier, 64 KB
// from L2 / HBM into LDS, barrier; the coefficient vectors) add another 120 k: 120 us in isolation, 135 us inside the step (cold
// feature chunks) against 62 + 56 us for the two ProbDZ launches it would replace, whose 2 048 workgroups hide exactly that latency.
#pragma once
#include "../../adkf_ift_amd/csrc/hyper.h"

namespace adkf {

constexpr int DZ_LD = HY_LDM;                                        // 144: conflict-free along rows and down columns
constexpr int DZ_LDS_FLOATS = 2 * HY_BUF + 3 * HY_N + 1024;          // A image, B image, coef_s, coef_q, ones, scratch
constexpr size_t DZ_LDS_BYTES = sizeof(float) * DZ_LDS_FLOATS;

struct DzArgs {
    const float *Wss, *Wqs, *Wqq, *Zs, *Zq;
    float *dZs, *dZq;
    int d, T;
};

// out[i] (+)= sum_k M[i][k]: thread (row, quarter), 16-byte groups interleaved over the quarters
__device__ __forceinline__ void dz_rowsum(const float* M, float* out, bool accumulate) {
    const int tid = threadIdx.x, i = tid >> 2, q = tid & 3;
    float s = 0.f;
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const float4 c4 = *reinterpret_cast<const float4*>(M + i * DZ_LD + 16 * u + 4 * q);
        s += (c4.x + c4.y) + (c4.z + c4.w);
    }
    s += dpp_f<DPP_XOR1>(s);
    s += dpp_f<DPP_XOR2>(s);
    if (q == 0) out[i] = accumulate ? out[i] + s : s;
}

// image[r][c] = scale * src[r][c], a 128 x 128 block with 16-byte accesses - in two halves, so that the loads of the NEXT image
// are in flight while the current product runs (dz_fetch before it, dz_put behind the barrier that follows it)
__device__ __forceinline__ void dz_fetch(float4 (&v)[8], const float* src, int src_ld) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const float4*>(src + (size_t)((tid >> 5) + 16 * u) * src_ld + (tid & 31) * 4);
}
__device__ __forceinline__ void dz_put(float* buf, const float4 (&v)[8], float scale) {
    const int tid = threadIdx.x;
#pragma unroll
    for (int u = 0; u < 8; ++u)
        *reinterpret_cast<float4*>(buf + ((tid >> 5) + 16 * u) * DZ_LD + (tid & 31) * 4) =
            make_float4(scale * v[u].x, scale * v[u].y, scale * v[u].z, scale * v[u].w);
}
__device__ __forceinline__ void dz_image(float* buf, const float* src, int src_ld, float scale) {
    float4 v[8];
    dz_fetch(v, src, src_ld);
    dz_put(buf, v, scale);
}

// dZ[i][c0 + j] = coef[i] Z[i][c0 + j] - acc   in the accumulator layout (i = 16 w + 4 g + y, j = 16 x + p)
__device__ __forceinline__ void dz_out(const f32x4_t (&acc)[8], const float* coef, const float* Z, float* dZ, int d, int c0) {
    const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4, w = threadIdx.x >> 6, i0 = 16 * w + 4 * g;
    float cf[4];
#pragma unroll
    for (int y = 0; y < 4; ++y) cf[y] = coef[i0 + y];
    float zv[8][4];
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) zv[x][y] = Z[(size_t)(i0 + y) * d + c0 + 16 * x + p];
#pragma unroll
    for (int x = 0; x < 8; ++x)
#pragma unroll
        for (int y = 0; y < 4; ++y) dZ[(size_t)(i0 + y) * d + c0 + 16 * x + p] = fmaf(cf[y], zv[x][y], -acc[x][y]);
}

__global__ __launch_bounds__(HY_NT, 1) void k_dz(DzArgs a) {
    extern __shared__ __align__(16) float dz_lds[];
    int t, tile;
    if (!task_tile(a.T, 1, t, tile)) return;
    const int tid = threadIdx.x, d = a.d;
    float* X = dz_lds;                 // weight image
    float* Y = X + HY_BUF;             // feature chunk [k][c]
    float* cs = Y + HY_BUF;            // coef_s
    float* cq = cs + HY_N;             // coef_q
    float* ones = cq + HY_N;
    float* scratch = ones + HY_N;      // 1024 floats
    const float* Wss = a.Wss + (size_t)t * HY_N * HY_N;
    const float* Wqs = a.Wqs + (size_t)t * HY_N * HY_N;
    const float* Wqq = a.Wqq + (size_t)t * HY_N * HY_N;
    const float* Zs = a.Zs + (size_t)t * HY_N * d;
    const float* Zq = a.Zq + (size_t)t * HY_N * d;
    float* dZs = a.dZs ? a.dZs + (size_t)t * HY_N * d : nullptr;
    float* dZq = a.dZq ? a.dZq + (size_t)t * HY_N * d : nullptr;
    if (tid < HY_N) ones[tid] = 1.f;

    // Per pass of 256 feature columns (L | R): first dZs (accumulators sL, sR), then dZq (qL, qR) - two sets at a time keep the
    // fragments double-buffered without spilling.  Every image but the very first is fetched into registers before the product in
    // front of it and stored behind the barrier that ends that product.
    for (int c0 = 0; c0 < d; c0 += 2 * HY_N) {
        const bool two = c0 + HY_N < d;        // a second chunk of 128 columns in this pass (uniform)
        const bool first = c0 == 0;
        const int cL = c0, cR = two ? c0 + HY_N : c0;
        float4 pf[8];
        f32x4_t aL[8], aR[8];
        // ---------------- dZs = coef_s . Zs - 4 Wss Zs - (2 Wqs)^T Zq ----------------
        hy_zero(aL); hy_zero(aR);
        __syncthreads();                                            // (the previous pass is done with both images)
        dz_image(X, Wss, HY_N, 4.f);
        dz_image(Y, Zs + cL, d, 1.f);
        __syncthreads();
        if (first) dz_rowsum(X, cs, false);
        dz_fetch(pf, Zs + cR, d);
        ADKF_SST(0);
        hy_gemm<false, true>(aL, X, DZ_LD, Y, DZ_LD, 8);            // 4 Wss Zs[:, L]
        ADKF_SST(1);
        __syncthreads();
        dz_put(Y, pf, 1.f);
        __syncthreads();
        dz_fetch(pf, Wqs, HY_N);
        ADKF_SST(2);
        if (two) hy_gemm<false, true>(aR, X, DZ_LD, Y, DZ_LD, 8);   // 4 Wss Zs[:, R]
        ADKF_SST(3);
        __syncthreads();
        dz_put(X, pf, 2.f);
        dz_image(Y, Zq + cR, d, 1.f);
        __syncthreads();
        if (first) {
            dz_rowsum(X, cq, false);
            hy_colsum<false>(X, DZ_LD, nullptr, 0, HY_N, ones, scratch, scratch + 512, nullptr);   // colsum(2 Wqs) (barriers inside)
            if (tid < HY_N) cs[tid] += scratch[512 + tid];
        }
        dz_fetch(pf, Zq + cL, d);
        ADKF_SST(4);
        if (two) hy_gemm<true, true>(aR, X, DZ_LD, Y, DZ_LD, 8);    // (2 Wqs)^T Zq[:, R]
        ADKF_SST(5);
        __syncthreads();
        dz_put(Y, pf, 1.f);
        __syncthreads();
        dz_fetch(pf, Zs + cL, d);
        ADKF_SST(6);
        hy_gemm<true, true>(aL, X, DZ_LD, Y, DZ_LD, 8);             // (2 Wqs)^T Zq[:, L]
        ADKF_SST(7);
        __syncthreads();                                            // (coef_s complete and visible)
        ADKF_SST(8);
        if (dZs) { dz_out(aL, cs, Zs, dZs, d, cL); if (two) dz_out(aR, cs, Zs, dZs, d, cR); }
        ADKF_SST(9);
        // ---------------- dZq = coef_q . Zq - 2 Wqs Zs - 4 Wqq Zq   (X still holds 2 Wqs) ----------------
        hy_zero(aL); hy_zero(aR);
        dz_put(Y, pf, 1.f);
        __syncthreads();
        dz_fetch(pf, Zs + cR, d);
        ADKF_SST(10);
        hy_gemm<false, true>(aL, X, DZ_LD, Y, DZ_LD, 8);            // 2 Wqs Zs[:, L]
        ADKF_SST(11);
        __syncthreads();
        dz_put(Y, pf, 1.f);
        __syncthreads();
        dz_fetch(pf, Wqq, HY_N);
        ADKF_SST(12);
        if (two) hy_gemm<false, true>(aR, X, DZ_LD, Y, DZ_LD, 8);   // 2 Wqs Zs[:, R]
        ADKF_SST(13);
        __syncthreads();
        dz_put(X, pf, 4.f);
        dz_image(Y, Zq + cR, d, 1.f);
        __syncthreads();
        if (first) dz_rowsum(X, cq, true);
        dz_fetch(pf, Zq + cL, d);
        ADKF_SST(14);
        if (two) hy_gemm<false, true>(aR, X, DZ_LD, Y, DZ_LD, 8);   // 4 Wqq Zq[:, R]
        ADKF_SST(15);
        __syncthreads();
        dz_put(Y, pf, 1.f);
        __syncthreads();
        ADKF_SST(16);
        hy_gemm<false, true>(aL, X, DZ_LD, Y, DZ_LD, 8);            // 4 Wqq Zq[:, L]
        ADKF_SST(17);
        __syncthreads();                                            // (coef_q complete and visible)
        ADKF_SST(18);
        if (dZq) { dz_out(aL, cq, Zq, dZq, d, cL); if (two) dz_out(aR, cq, Zq, dZq, d, cR); }
        ADKF_SST(19);
    }
}

}  // namespace adkf
