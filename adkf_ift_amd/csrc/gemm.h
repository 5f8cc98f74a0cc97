// Batched small-matrix GEMM on the fp32-input MFMA (v_mfma_f32_16x16x4_f32), one 64x64 output tile per
// 256-thread workgroup, operands produced ON THE FLY by a problem functor (kernel-matrix entries from
// squared distances, Omega from S^-1 and e, ...) and a fused epilogue functor (elementwise chain rule +
// reductions), so no intermediate kernel matrix is ever written to HBM.
//
// A problem type P provides
//   static constexpr bool A_KCONTIG / B_KCONTIG : is the operand contiguous in memory along k?  (chooses
//                                                  the coalesced thread->element map and the LDS layout)
//   static constexpr int  NRED                   : per-tile reductions the epilogue accumulates
//   __device__ bool  setup(int task)             : loads per-task sizes/scalars; false = nothing to do
//   __device__ int   M(), N(), K()               : per-task logical sizes
//   __device__ float a(int i, int k), b(int k, int j)   : operand entries (callers guarantee in-range)
//   __device__ void  epi(int i, int j, float acc, float* red)
//   __device__ void  store_red(int tile, const float* red)   (only when NRED > 0; called by thread 0)
#pragma once
#include "device_utils.h"

namespace adkf {

constexpr int GT = 64;        // tile edge
constexpr int GK = 16;        // k chunk
constexpr int LD_MN = GK + 1; // [mn][k] layout, K-contiguous operands (conflict-free b32 fragment reads)
constexpr int LD_K = GT + 16; // [k][mn] layout, MN-contiguous operands (LD % 32 == 16)

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <class P>
__global__ __launch_bounds__(256) void k_bgemm(P p, int T, int tiles_m, int tiles_n) {
    int task, tile;
    if (!task_tile(T, tiles_m * tiles_n, task, tile)) return;
    if (!p.setup(task)) return;
    const int M = p.M(), N = p.N(), K = p.K();
    const int m0 = (tile / tiles_n) * GT, n0 = (tile % tiles_n) * GT;
    if (m0 >= M || n0 >= N) {  // tile outside this (ragged) task: contributes zero partials
        if (P::NRED > 0 && threadIdx.x == 0) {
            float z[(P::NRED > 0 ? P::NRED : 1)];
            for (int q = 0; q < (P::NRED > 0 ? P::NRED : 1); ++q) z[q] = 0.f;
            p.store_red(tile, z);
        }
        return;
    }

    __shared__ float As[(P::A_KCONTIG ? GT * LD_MN : GK * LD_K)];
    __shared__ float Bs[(P::B_KCONTIG ? GT * LD_MN : GK * LD_K)];
    __shared__ float red_s[(P::NRED > 0 ? P::NRED * 4 : 1)];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 1, wc = wv & 1;  // 2x2 waves, 32x32 each
    const int fi = lane & 15, fk = lane >> 4;

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    for (int k0 = 0; k0 < K; k0 += GK) {
        // ---- stage A tile (64 x 16) ----
        if (P::A_KCONTIG) {
            const int kk = tid & 15, r0 = tid >> 4;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int r = r0 + ps * 16;
                const int gi = m0 + r, gk = k0 + kk;
                As[r * LD_MN + kk] = (gi < M && gk < K) ? p.a(gi, gk) : 0.f;
            }
        } else {
            const int mm = tid & 63, kq = tid >> 6;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int kk = kq + ps * 4;
                const int gi = m0 + mm, gk = k0 + kk;
                As[kk * LD_K + mm] = (gi < M && gk < K) ? p.a(gi, gk) : 0.f;
            }
        }
        // ---- stage B tile (16 x 64) ----
        if (P::B_KCONTIG) {
            const int kk = tid & 15, r0 = tid >> 4;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int r = r0 + ps * 16;
                const int gj = n0 + r, gk = k0 + kk;
                Bs[r * LD_MN + kk] = (gj < N && gk < K) ? p.b(gk, gj) : 0.f;
            }
        } else {
            const int nn = tid & 63, kq = tid >> 6;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int kk = kq + ps * 4;
                const int gj = n0 + nn, gk = k0 + kk;
                Bs[kk * LD_K + nn] = (gj < N && gk < K) ? p.b(gk, gj) : 0.f;
            }
        }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < GK / 4; ++s) {
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int r = wr * 32 + i * 16 + fi;
                af[i] = P::A_KCONTIG ? As[r * LD_MN + 4 * s + fk] : As[(4 * s + fk) * LD_K + r];
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c = wc * 32 + j * 16 + fi;
                bf[j] = P::B_KCONTIG ? Bs[c * LD_MN + 4 * s + fk] : Bs[(4 * s + fk) * LD_K + c];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // ---- epilogue: C/D map of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg ----
    float red[(P::NRED > 0 ? P::NRED : 1)];
#pragma unroll
    for (int q = 0; q < (P::NRED > 0 ? P::NRED : 1); ++q) red[q] = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = m0 + wr * 32 + i * 16 + fk * 4 + r;
                const int gj = n0 + wc * 32 + j * 16 + fi;
                if (gi < M && gj < N) p.epi(gi, gj, acc[i][j][r], red);
            }
    if (P::NRED > 0) {
        block_sum<(P::NRED > 0 ? P::NRED : 1), 256>(red, red_s);
        if (tid == 0) p.store_red(tile, red);
    }
}

}  // namespace adkf
