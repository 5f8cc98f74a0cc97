// Harness row H (fs_mol/utils/adaptive_dkt_utils.py:402-413): task-mean, clip-by-global-norm and the Adam update of the
// outer parameters as two launches per tensor instead of torch's norm + clamp + mul + fused-Adam chain (whose
// multi_tensor_apply gives a whole 64 K-element chunk to ONE workgroup: 43 us for a 256 x 256 parameter).
//
//   k_grad_sumsq     : partials[b] = sum over block b's grid-stride slice of g^2            (SUMSQ_PARTS blocks, fixed)
//   k_clip_adam      : every block re-adds the partials IN INDEX ORDER (deterministic, identical on every rank, so
//                      data-parallel replicas stay bit-identical), forms
//                          coef = scale * min(1, clip / (scale * |g| + 1e-6))               (clip_grad_norm_)
//                      and applies torch.optim.Adam's update (no amsgrad, L2 weight decay) to its slice; the clipped
//                      gradient is written back so that p.grad reads as it would after the torch sequence.
#pragma once
#include "device_utils.h"

namespace adkf {

constexpr int SUMSQ_PARTS = 256;   // == ADKF_SUMSQ_PARTS
constexpr int STEP_NT = 256;

__global__ __launch_bounds__(STEP_NT) void k_grad_sumsq(const float* __restrict__ g, long n, float* __restrict__ partials) {
    __shared__ float red[STEP_NT / 64];
    float s = 0.f;
    const long n4 = n >> 2;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (long i = (long)blockIdx.x * STEP_NT + threadIdx.x; i < n4; i += (long)SUMSQ_PARTS * STEP_NT) {
        float4 x = g4[i];
        s += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
        float x = g[(n4 << 2) + threadIdx.x];
        s += x * x;
    }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < STEP_NT / 64; ++w) t += red[w];
        partials[blockIdx.x] = t;
    }
}

struct AdamArgs {
    float* p;
    float* g;
    float* m;
    float* v;
    long n;
    const float* partials;
    int n_partials;
    float scale, clip, step_size, omb1, beta2, omb2, eps, weight_decay, bias2_sqrt;   // host doubles rounded once, as torch does
};

__device__ __forceinline__ void adam_one(float& p, float& g, float& m, float& v, float coef, const AdamArgs& a) {
    g *= coef;
    float gg = g + a.weight_decay * p;
    m = m + (gg - m) * a.omb1;
    v = a.beta2 * v + a.omb2 * gg * gg;
    float denom = sqrtf(v) / a.bias2_sqrt + a.eps;
    p -= a.step_size * (m / denom);
}

__global__ __launch_bounds__(STEP_NT) void k_clip_adam(AdamArgs a) {
    __shared__ float s_coef;
    if (threadIdx.x < 64) {
        // fixed order: lane l adds partials l, l+64, ... then the 64 lane sums go through the same butterfly everywhere
        float t = 0.f;
        for (int i = threadIdx.x; i < a.n_partials; i += 64) t += a.partials[i];
        t = wave_sum(t);
        if (threadIdx.x == 0) {
            float norm = sqrtf(t);
            s_coef = a.scale * fminf(1.f, a.clip / (a.scale * norm + 1e-6f));
        }
    }
    __syncthreads();
    const float coef = s_coef;
    const long n4 = a.n >> 2;
    float4* p4 = reinterpret_cast<float4*>(a.p);
    float4* g4 = reinterpret_cast<float4*>(a.g);
    float4* m4 = reinterpret_cast<float4*>(a.m);
    float4* v4 = reinterpret_cast<float4*>(a.v);
    for (long i = (long)blockIdx.x * STEP_NT + threadIdx.x; i < n4; i += (long)gridDim.x * STEP_NT) {
        float4 p = p4[i], g = g4[i], m = m4[i], v = v4[i];
        adam_one(p.x, g.x, m.x, v.x, coef, a);
        adam_one(p.y, g.y, m.y, v.y, coef, a);
        adam_one(p.z, g.z, m.z, v.z, coef, a);
        adam_one(p.w, g.w, m.w, v.w, coef, a);
        p4[i] = p; g4[i] = g; m4[i] = m; v4[i] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        long i = (n4 << 2) + threadIdx.x;
        adam_one(a.p[i], a.g[i], a.m[i], a.v[i], coef, a);
    }
}

// One launch for ONE small tensor (n <= CLIP_ADAM_ONE_MAX): every workgroup re-adds the squares of the WHOLE gradient in the same fixed
// order (thread t takes the 16-byte groups t, t + 1024, ...; wave sums; the sixteen wave totals in index order) - identical bits in every
// workgroup and on every rank -, then clips and updates its own slice like k_clip_adam.  Optionally (planes_t != null, the tensor a
// row-major [K][N] weight as x @ w takes it, K and N multiples of 64) the three bfloat16 pieces of every NEW weight go straight into the transposed planes
// [3][N][K] that adkf_dense_forward wants for y = x w (dense_x3.h: k_split3_t's output, bit for bit): the weight is split where it is
// updated instead of by a launch of its own in the next forward pass.  2 (+ 1) launches of the C2 step become one.
constexpr long CLIP_ADAM_ONE_MAX = 131072;
constexpr int STEP1_NT = 1024;

struct AdamOneArgs {
    AdamArgs a;                 // (partials / n_partials unused)
    unsigned short* planes_t;   // or null
    int K, N;
};

__device__ __forceinline__ void split_one(float x, unsigned short& q0, unsigned short& q1, unsigned short& q2);   // dense_x3.h

constexpr int STEP1_TILE = 64, STEP1_LD = STEP1_TILE + 2;   // planes: a workgroup owns 64 k x 64 n weights; LDS rows of 33 words

__global__ __launch_bounds__(STEP1_NT) void k_clip_adam_one(AdamOneArgs o) {
    const AdamArgs& a = o.a;
    __shared__ float red[STEP1_NT / 64];
    __shared__ float s_coef;
    __shared__ unsigned short tile[3][STEP1_TILE][STEP1_LD];   // [piece][n][k]
    const long n4 = a.n >> 2;
    const float4* gq = reinterpret_cast<const float4*>(a.g);
    float s = 0.f;
    long i = threadIdx.x;
    for (; i + 7 * STEP1_NT < n4; i += 8 * STEP1_NT) {   // eight loads in flight; added in index order (the order is part of the result)
        float4 x[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) x[u] = gq[i + u * STEP1_NT];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += x[u].x * x[u].x + x[u].y * x[u].y + x[u].z * x[u].z + x[u].w * x[u].w;
    }
    for (; i < n4; i += STEP1_NT) {
        const float4 x = gq[i];
        s += x.x * x.x + x.y * x.y + x.z * x.z + x.w * x.w;
    }
    if (threadIdx.x < (a.n & 3)) { const float x = a.g[(n4 << 2) + threadIdx.x]; s += x * x; }
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int w = 0; w < STEP1_NT / 64; ++w) t += red[w];
        s_coef = a.scale * fminf(1.f, a.clip / (a.scale * sqrtf(t) + 1e-6f));
    }
    __syncthreads();
    const float coef = s_coef;
    float4* p4 = reinterpret_cast<float4*>(a.p);
    float4* g4 = reinterpret_cast<float4*>(a.g);
    float4* m4 = reinterpret_cast<float4*>(a.m);
    float4* v4 = reinterpret_cast<float4*>(a.v);
    if (o.planes_t) {
        // (host: K and N multiples of 64, grid = (K / 64) * (N / 64)) lane t takes w[k0 + t / 16][n0 + 4 (t % 16) .. + 3]: rows of w are
        // read 256 bytes at a time; the pieces go through LDS so that the planes [n][k] are written 128 bytes of k at a time
        const int tiles_n = o.N / STEP1_TILE;
        const int k0 = ((int)blockIdx.x / tiles_n) * STEP1_TILE, n0 = ((int)blockIdx.x % tiles_n) * STEP1_TILE;
        const int kl = threadIdx.x >> 4, nl = (threadIdx.x & 15) << 2;
        const long at4 = ((long)(k0 + kl) * o.N + n0 + nl) >> 2;
        float4 p = p4[at4], g = g4[at4], m = m4[at4], v = v4[at4];
        adam_one(p.x, g.x, m.x, v.x, coef, a);
        adam_one(p.y, g.y, m.y, v.y, coef, a);
        adam_one(p.z, g.z, m.z, v.z, coef, a);
        adam_one(p.w, g.w, m.w, v.w, coef, a);
        p4[at4] = p; g4[at4] = g; m4[at4] = m; v4[at4] = v;
        const float w4[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
        for (int x = 0; x < 4; ++x) split_one(w4[x], tile[0][nl + x][kl], tile[1][nl + x][kl], tile[2][nl + x][kl]);
        __syncthreads();
        const size_t plane_words = (size_t)o.N * o.K / 2;
        uint32_t* out = reinterpret_cast<uint32_t*>(o.planes_t);
#pragma unroll
        for (int j = 0; j < 3 * STEP1_TILE * (STEP1_TILE / 2) / STEP1_NT; ++j) {
            const int w = threadIdx.x + j * STEP1_NT;
            const int q = w / (STEP1_TILE * STEP1_TILE / 2), r = (w / (STEP1_TILE / 2)) % STEP1_TILE, c = w % (STEP1_TILE / 2);
            const uint32_t val = *reinterpret_cast<const uint32_t*>(&tile[q][r][2 * c]);
            out[q * plane_words + (((size_t)(n0 + r) * o.K + k0) >> 1) + c] = val;
        }
        return;
    }
    for (long i4 = (long)blockIdx.x * STEP1_NT + threadIdx.x; i4 < n4; i4 += (long)gridDim.x * STEP1_NT) {
        float4 p = p4[i4], g = g4[i4], m = m4[i4], v = v4[i4];
        adam_one(p.x, g.x, m.x, v.x, coef, a);
        adam_one(p.y, g.y, m.y, v.y, coef, a);
        adam_one(p.z, g.z, m.z, v.z, coef, a);
        adam_one(p.w, g.w, m.w, v.w, coef, a);
        p4[i4] = p; g4[i4] = g; m4[i4] = m; v4[i4] = v;
    }
    if (blockIdx.x == 0 && threadIdx.x < (a.n & 3)) {
        long e = (n4 << 2) + threadIdx.x;
        adam_one(a.p[e], a.g[e], a.m[e], a.v[e], coef, a);
    }
}

}  // namespace adkf
