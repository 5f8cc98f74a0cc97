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

}  // namespace adkf
