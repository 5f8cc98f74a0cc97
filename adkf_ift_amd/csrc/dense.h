// Dense fp32 layers of the feature extractor (BOOM 128 <-> 1024, message-output projection, read-out MLPs, the fc head
// 2560 -> 2048 -> 2048; fs_mol/modules/gnn.py:95,497-513, graph_readout.py:119-177, models/adaptive_dkt.py:61-65) on the
// same fp32 MFMA tile kernel as the GP stages (gemm.h).  Round 1 left them to torch / hipBLASLt, whose fp32 kernels ran at
// 5 .. 30 TFLOP/s on these shapes (profiles/r01_c3_kernel_stats.csv: 3.5 ms for 2304 x 2048 x 2048) - 45 of the 85 ms of a
// C3 step.
//     forward    Y = act(X W^T + b)                     X [M, K], W [N, K], Y [M, N];  act: none | relu | leaky_relu(0.01)
//     backward   G = dY . act'(Y)   (the activation mask is applied while the operand is loaded: act' depends on sign(Y) only)
//                dX = G W            dW += G^T X  (split over row chunks, atomic accumulation)        db += colsum(G)
// A k_bgemm "task" is one 64-row panel of the output (x one chunk of the reduction for dW), so the XCD-aware block map
// of device_utils.h spreads the panels over the eight XCDs and every block has work.
#pragma once
#include "problems.h"

namespace adkf {

enum { DENSE_ACT_NONE = 0, DENSE_ACT_RELU = 1, DENSE_ACT_LEAKY = 2 };

__device__ __forceinline__ float dense_act(float v, int act) {
    return act == DENSE_ACT_RELU ? fmaxf(v, 0.f) : (act == DENSE_ACT_LEAKY ? (v > 0.f ? v : 0.01f * v) : v);
}
__device__ __forceinline__ float dense_dact(float y, int act) {   // derivative from the ACTIVATED value (same sign as the input)
    return act == DENSE_ACT_RELU ? (y > 0.f ? 1.f : 0.f) : (act == DENSE_ACT_LEAKY ? (y > 0.f ? 1.f : 0.01f) : 1.f);
}

// Y = act(X W^T + b): A = X (k contiguous), B[k][j] = W[j][k] (k contiguous)
struct ProbDenseFwd {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = true;
    static constexpr int NRED = 0;
    const float *X, *W, *bias; float* Y; int M_, N_, K_, act; bool vec;
    int m_base, rows;
    __device__ bool setup(int t) { m_base = t * GT; rows = min(GT, M_ - m_base); return rows > 0; }
    __device__ int M() const { return rows; } __device__ int N() const { return N_; } __device__ int K() const { return K_; }
    __device__ float a(int i, int k) const { return X[(size_t)(m_base + i) * K_ + k]; }
    __device__ float b(int k, int j) const { return W[(size_t)j * K_ + k]; }
    __device__ void a4(int i, int k, float (&v)[4]) const { ld4(X + (size_t)(m_base + i) * K_ + k, v); }
    __device__ void b4(int k, int j, float (&v)[4]) const { ld4(W + (size_t)j * K_ + k, v); }
    __device__ void epi(int i, int j, float acc, float*) const {
        Y[(size_t)(m_base + i) * N_ + j] = dense_act(acc + (bias ? bias[j] : 0.f), act);
    }
    __device__ void store_red(int, const float*) const {}
};

// dX = G W,  G = dY . act'(Y): A = G (n contiguous = the reduction index), B[n][j] = W[n][j] (j contiguous)
struct ProbDenseBwdX {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = false;
    static constexpr int NRED = 0;
    const float *dY, *Yact, *W; float* dX; int M_, N_, K_, act, accumulate; bool vec;
    int m_base, rows;
    __device__ bool setup(int t) { m_base = t * GT; rows = min(GT, M_ - m_base); return rows > 0; }
    __device__ int M() const { return rows; } __device__ int N() const { return K_; } __device__ int K() const { return N_; }
    __device__ float g(size_t o) const { return act ? dY[o] * dense_dact(Yact[o], act) : dY[o]; }
    __device__ float a(int i, int k) const { return g((size_t)(m_base + i) * N_ + k); }
    __device__ float b(int k, int j) const { return W[(size_t)k * K_ + j]; }
    __device__ void a4(int i, int k, float (&v)[4]) const {
        const size_t o = (size_t)(m_base + i) * N_ + k;
        ld4(dY + o, v);
        if (act) {
            float y[4]; ld4(Yact + o, y);
#pragma unroll
            for (int x = 0; x < 4; ++x) v[x] *= dense_dact(y[x], act);
        }
    }
    __device__ void b4(int k, int j, float (&v)[4]) const { ld4(W + (size_t)k * K_ + j, v); }
    __device__ void epi(int i, int j, float acc, float*) const {
        float* p = dX + (size_t)(m_base + i) * K_ + j;
        *p = accumulate ? *p + acc : acc;
    }
    __device__ void store_red(int, const float*) const {}
};

// dW += G^T X over one chunk of rows: task = (chunk, 64-row panel of dW); A[n][m] = G[m][n] (n contiguous), B[m][j] = X[m][j]
struct ProbDenseBwdW {
    static constexpr bool A_KCONTIG = false, B_KCONTIG = false;
    static constexpr int NRED = 0;
    const float *dY, *Yact, *X; float* dW; int M_, N_, K_, act, chunk, panels; bool vec;
    int n_base, rows, m0, mlen;
    __device__ bool setup(int t) {
        const int s = t / panels, pn = t - s * panels;
        n_base = pn * GT; rows = min(GT, N_ - n_base);
        m0 = s * chunk; mlen = min(chunk, M_ - m0);
        return rows > 0 && mlen > 0;
    }
    __device__ int M() const { return rows; } __device__ int N() const { return K_; } __device__ int K() const { return mlen; }
    __device__ float g(size_t o) const { return act ? dY[o] * dense_dact(Yact[o], act) : dY[o]; }
    __device__ float a(int i, int k) const { return g((size_t)(m0 + k) * N_ + n_base + i); }
    __device__ float b(int k, int j) const { return X[(size_t)(m0 + k) * K_ + j]; }
    __device__ void a4(int i, int k, float (&v)[4]) const {
        const size_t o = (size_t)(m0 + k) * N_ + n_base + i;
        ld4(dY + o, v);
        if (act) {
            float y[4]; ld4(Yact + o, y);
#pragma unroll
            for (int x = 0; x < 4; ++x) v[x] *= dense_dact(y[x], act);
        }
    }
    __device__ void b4(int k, int j, float (&v)[4]) const { ld4(X + (size_t)(m0 + k) * K_ + j, v); }
    __device__ void epi(int i, int j, float acc, float*) const { atomicAdd(dW + (size_t)(n_base + i) * K_ + j, acc); }
    __device__ void store_red(int, const float*) const {}
};

// db[j] += sum_m G[m][j]: 64 columns x 4 row groups per workgroup over a chunk of 256 rows, atomically accumulated
struct DenseDbArgs { const float *dY, *Yact; float* db; int M_, N_, act; };
__global__ __launch_bounds__(256) void k_dense_dbias(DenseDbArgs a) {
    __shared__ float part[4][64];
    const int cl = threadIdx.x & 63, gq = threadIdx.x >> 6, c = blockIdx.x * 64 + cl;
    const int r0 = blockIdx.y * 256, r1 = min(a.M_, r0 + 256);
    float s = 0.f;
    if (c < a.N_)
        for (int r = r0 + gq; r < r1; r += 4) {
            const size_t o = (size_t)r * a.N_ + c;
            s += a.act ? a.dY[o] * dense_dact(a.Yact[o], a.act) : a.dY[o];
        }
    part[gq][cl] = s;
    __syncthreads();
    if (gq == 0 && c < a.N_) atomicAdd(a.db + c, (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]));
}

}  // namespace adkf
