// Device-side helpers shared by every kernel of libadkf_gp (gfx950 only: wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace adkf {

constexpr int WAVE = 64;
constexpr float NOISE_LB = 1e-4f;       // gpytorch GaussianLikelihood GreaterThan(1e-4)
constexpr float LOG_2PI = 1.8378770664093453f;
constexpr float SQRT5 = 2.23606797749979f;

// ---------------------------------------------------------------------------------------------------
// per-task scalar slots (float) in the workspace
// ---------------------------------------------------------------------------------------------------
enum Scal : int {
    S_NOISE = 0, S_OS, S_LS,            // transformed hyper-parameters
    S_D1N, S_D1S, S_D1L,                // d transformed / d raw   (sigmoid)
    S_D2N, S_D2S, S_D2L,                // second derivative
    S_FIN, S_GIN0, S_GIN1, S_GIN2,      // f_inner and its raw gradient
    S_GT0, S_GT1, S_GT2,                // d (nll - log priors) / d transformed
    S_LOGDET, S_TRAINV, S_AA, S_YA, S_TRAINVG, S_AGA,
    S_H0, S_H1, S_H2, S_H3, S_H4, S_H5, S_H6, S_H7, S_H8,
    S_GOUT0, S_GOUT1, S_GOUT2,
    S_V0, S_V1, S_V2,
    S_CN, S_CS, S_CL,                   // coefficients of B_v = cn I + cs K + cl G
    S_FOUT, S_LOGDETS,
    S_QQ_TR, S_QQ_K, S_QQ_L,            // reductions of the W_qq kernel: tr(Om), <Om,kappa>, <Om,dK/dl>
    S_PIVR_A, S_PIVR_S,                 // pivot ratio max d_k / min d_k of the sweeps of A and Sigma_q (refine64.h)
    S_CONDA,                            // (s + noise) max_i (A^-1)_ii: who takes the float32 refinement of C and alpha (problems.h ProbCres)
    S_AREF,                             // 1 once k_alpha_refine has refined THIS alpha (the step must not be applied twice: predict followed by
                                        // ift_hypergrad on one batch with REUSE_INNER); cleared by whoever writes a new alpha
    S_COUNT_ = 64
};
constexpr int NSCAL = 64;

// per-task vector slots (each nvec_ld floats)
enum Vec : int { V_ALPHA = 0, V_BETA, V_GAMMA, V_DELTA, V_W, V_R, V_E, V_CTE, V_MU, V_COUNT_ = 16 };
constexpr int NVEC = 16;

// ---------------------------------------------------------------------------------------------------
// math
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float softplus_f(float x) { return x > 20.f ? x : log1pf(expf(x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.f / (1.f + expf(-x)); }
__device__ __forceinline__ float inv_softplus_f(float y) { return y > 20.f ? y : y + logf(-expm1f(-y)); }

// exp for the kernel functions (argument <= 0): v_exp_f32 on a compensated product, 6 instructions and ~2 ulp
// instead of the ~15 of libm's expf (64 calls per lane per MLL evaluation).
__device__ __forceinline__ float exp_fast(float x) {
    const float L2E = 1.44269502162933349609375f, L2E_LO = 1.925963033e-8f;
    const float t = x * L2E;
    float e = fmaf(x, L2E, -t);
    e = fmaf(x, L2E_LO, e);
    const float r = __builtin_amdgcn_exp2f(t);
    return fmaf(r, e * 0.6931471805599453f, r);
}

// kappa(u), kappa'(u), kappa''(u); u = squared scaled distance  (oracle/closed_form.py::kappa)
// FAST selects exp_fast (used only inside the fit's line-search evaluations); everything that is reported or
// differentiated further uses libm's expf.
template <int KIND, bool FAST = false>
__device__ __forceinline__ void kappa3(float u, float& k0, float& k1, float& k2) {
    if (KIND == 0) {
        k0 = FAST ? exp_fast(-0.5f * u) : expf(-0.5f * u);
        k1 = -0.5f * k0;
        k2 = 0.25f * k0;
    } else {
        float r = sqrtf(u);
        float e = FAST ? exp_fast(-SQRT5 * r) : expf(-SQRT5 * r);
        k0 = (1.f + SQRT5 * r + (5.f / 3.f) * u) * e;
        k1 = -(5.f / 6.f) * (1.f + SQRT5 * r) * e;
        k2 = (25.f / 12.f) * e;
    }
}
__device__ __forceinline__ void kappa3(int kind, float u, float& k0, float& k1, float& k2) {
    if (kind == 0) kappa3<0>(u, k0, k1, k2); else kappa3<1>(u, k0, k1, k2);
}
template <int KIND, bool FAST = false>
__device__ __forceinline__ float kappa0_t(float u) {
    if (KIND == 0) return FAST ? exp_fast(-0.5f * u) : expf(-0.5f * u);
    const float r = sqrtf(u);
    return (1.f + SQRT5 * r + (5.f / 3.f) * u) * (FAST ? exp_fast(-SQRT5 * r) : expf(-SQRT5 * r));
}
__device__ __forceinline__ float kappa0(int kind, float u) {
    if (kind == 0) return expf(-0.5f * u);
    float r = sqrtf(u);
    return (1.f + SQRT5 * r + (5.f / 3.f) * u) * expf(-SQRT5 * r);
}

// ---------------------------------------------------------------------------------------------------
// reductions: wave shuffles first, one LDS hop across waves (deterministic order)
// ---------------------------------------------------------------------------------------------------
// Cross-lane moves inside a row of 16 lanes ride the VALU's DPP path (a few cycles) instead of the LDS crossbar
// (ds_bpermute: ~100+ cycles each, and __shfl_xor always compiles to it): xor-1 and xor-2 inside the quad, then the two
// mirrors.  After the four steps every lane holds its row's total; the four row totals are read as scalars.
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __int_as_float(dpp_i<CTRL>(__float_as_int(v))); }
constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E, DPP_HALF_MIRROR = 0x141, DPP_MIRROR = 0x140;

// Every lane of the wave must be active (the callers reduce in wave-uniform control flow); all lanes get the total.
__device__ __forceinline__ float wave_sum(float v) {
    v += dpp_f<DPP_XOR1>(v);
    v += dpp_f<DPP_XOR2>(v);
    v += dpp_f<DPP_HALF_MIRROR>(v);
    v += dpp_f<DPP_MIRROR>(v);
    const int b = __float_as_int(v);
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(b, 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(b, 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(b, 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(b, 48));
    return (r0 + r1) + (r2 + r3);
}
__device__ __forceinline__ int wave_sum_i(int v) {
    v += dpp_i<DPP_XOR1>(v);
    v += dpp_i<DPP_XOR2>(v);
    v += dpp_i<DPP_HALF_MIRROR>(v);
    v += dpp_i<DPP_MIRROR>(v);
    return (__builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16)) + (__builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48));
}
__device__ __forceinline__ float wave_max(float v) {
    v = fmaxf(v, dpp_f<DPP_XOR1>(v));
    v = fmaxf(v, dpp_f<DPP_XOR2>(v));
    v = fmaxf(v, dpp_f<DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dpp_f<DPP_MIRROR>(v));
    const int b = __float_as_int(v);
    return fmaxf(fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 0)), __int_as_float(__builtin_amdgcn_readlane(b, 16))),
                 fmaxf(__int_as_float(__builtin_amdgcn_readlane(b, 32)), __int_as_float(__builtin_amdgcn_readlane(b, 48))));
}
__device__ __forceinline__ float wave_min(float v) { return -wave_max(-v); }

// Sums K values over the whole block; every thread gets the totals.  `red` needs K * (NT/64) floats.
// Contains two barriers; safe to call repeatedly with the same scratch.
template <int K, int NT>
__device__ __forceinline__ void block_sum(float (&v)[K], float* red) {
    constexpr int NW = NT / WAVE;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int q = 0; q < K; ++q) v[q] = wave_sum(v[q]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int q = 0; q < K; ++q) red[q * NW + w] = v[q];
    }
    __syncthreads();
    // every thread sums the NW partials of each value, in wave order; they are fetched with 16-byte LDS reads when `red` allows
    // (K * NW dword reads by every lane were the larger part of this function: 56 at <7, 512>, 144 at <9, 1024>)
    if (NW % 4 == 0 && (reinterpret_cast<uintptr_t>(red) & 15u) == 0) {
#pragma unroll
        for (int q = 0; q < K; ++q) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NW / 4; ++i) {
                const float4 t = reinterpret_cast<const float4*>(red + q * NW)[i];
                s += t.x; s += t.y; s += t.z; s += t.w;
            }
            v[q] = s;
        }
    } else {
#pragma unroll
        for (int q = 0; q < K; ++q) {
            float s = 0.f;
#pragma unroll
            for (int i = 0; i < NW; ++i) s += red[q * NW + i];
            v[q] = s;
        }
    }
}

template <int NT>
__device__ __forceinline__ int block_sum_i(int v, int* red) {
    constexpr int NW = NT / WAVE;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_sum_i(v);
    __syncthreads();
    if (lane == 0) red[w] = v;
    __syncthreads();
    int s = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) s += red[i];
    return s;
}

// CB consecutive floats of one matrix row starting at column j0, of which the first n columns exist: 16-byte accesses where
// a group of four lies inside the row and `vec` says the row starts are 16-byte aligned (leading dimension a multiple of 4 on
// an aligned base), element-wise otherwise.  A thread's register block of a matrix is RB such segments: 8 instead of 32
// memory instructions per lane for the 4 x 8 blocks of the 128-point kernels.
template <int CB>
__device__ __forceinline__ void load_segment(const float* rowp, int j0, int n, bool row_ok, bool vec, float (&o)[CB]) {
    if constexpr (CB % 4 == 0) {
#pragma unroll
        for (int q = 0; q < CB / 4; ++q) {
            const int j = j0 + 4 * q;
            if (vec && row_ok && j + 4 <= n) {
                const float4 t = *reinterpret_cast<const float4*>(rowp + j);
                o[4 * q] = t.x; o[4 * q + 1] = t.y; o[4 * q + 2] = t.z; o[4 * q + 3] = t.w;
            } else {
#pragma unroll
                for (int x = 0; x < 4; ++x) o[4 * q + x] = (row_ok && j + x < n) ? rowp[j + x] : 0.f;
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < CB; ++c) o[c] = (row_ok && j0 + c < n) ? rowp[j0 + c] : 0.f;
    }
}
template <int CB>
__device__ __forceinline__ void store_segment(float* rowp, int j0, int n, bool row_ok, bool vec, const float (&v)[CB]) {
    if (!row_ok) return;
    if constexpr (CB % 4 == 0) {
#pragma unroll
        for (int q = 0; q < CB / 4; ++q) {
            const int j = j0 + 4 * q;
            if (vec && j + 4 <= n) {
                *reinterpret_cast<float4*>(rowp + j) = make_float4(v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]);
            } else {
#pragma unroll
                for (int x = 0; x < 4; ++x)
                    if (j + x < n) rowp[j + x] = v[4 * q + x];
            }
        }
    } else {
#pragma unroll
        for (int c = 0; c < CB; ++c)
            if (j0 + c < n) rowp[j0 + c] = v[c];
    }
}
__device__ __forceinline__ bool rows_aligned16(const void* base, int ld) { return ((reinterpret_cast<uintptr_t>(base) & 15u) == 0) && (ld & 3) == 0; }

// XCD-aware block -> (task, tile) map: consecutive block ids are dealt round-robin to the 8 XCDs, so all
// tiles of task t, in every kernel of the pipeline, run on the XCD labelled (t & 7) and find the task's
// matrices in that XCD's L2.  Speed only: any placement is correct.  Grid = roundup8(T) * tiles.
__device__ __forceinline__ bool task_tile(int T, int tiles, int& task, int& tile) {
    const int b = blockIdx.x;
    const int xcd = b & 7, idx = b >> 3;
    task = (idx / tiles) * 8 + xcd;
    tile = idx % tiles;
    return task < T;
}

}  // namespace adkf
