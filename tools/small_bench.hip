// Phase stamps and launch time of k_hess at the C2 shape (256 tasks x 128 points).  Not part of the library.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -DADKF_STAMP_SMALL=1 -I adkf_ift_amd/csrc tools/small_bench.hip -o tools/small_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "kernels.h"
using namespace adkf;

int main() {
    const int T = 256, n = 128;
    std::vector<float> h((size_t)T * n * n);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f; }
    float *Ainv, *P, *D2, *ys, *pri, *scal, *vecs;
    hipMalloc(&Ainv, h.size() * 4); hipMalloc(&P, h.size() * 4); hipMalloc(&D2, h.size() * 4);
    hipMalloc(&ys, (size_t)T * n * 4); hipMalloc(&pri, T * 16); hipMalloc(&scal, (size_t)T * NSCAL * 4); hipMalloc(&vecs, (size_t)T * NVEC * n * 4);
    hipMemcpy(Ainv, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(P, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(D2, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> one((size_t)T * NSCAL, 1.f);
    hipMemcpy(scal, one.data(), one.size() * 4, hipMemcpyHostToDevice);
    hipMemset(vecs, 0, (size_t)T * NVEC * n * 4); hipMemset(ys, 0, (size_t)T * n * 4);
    std::vector<float> pr(T * 4, 0.25f);
    hipMemcpy(pri, pr.data(), pr.size() * 4, hipMemcpyHostToDevice);
    HessArgs a;
    a.tv.n_s = nullptr; a.tv.n_q = nullptr; a.tv.ns_ld = n; a.tv.nq_ld = n; a.tv.vld = n; a.tv.kind = 0; a.tv.scal = scal; a.tv.vecs = vecs; a.tv.vec = true;
    a.Ainv = Ainv; a.P = P; a.D2ss = D2; a.y_s = ys; a.priors = pri; a.scal = scal; a.vecs = vecs; a.T = T;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) k_hess<<<T, SMALL_NT>>>(a);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("k_hess %.2f us per launch\n", ms * 1000 / 20);
    }
#if ADKF_STAMP_SMALL
    unsigned long long st[16];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_small_stamps), sizeof(st));
    for (int k = 1; k <= 4; ++k) printf("k_hess phase %d: %llu cycles\n", k, st[k] - st[k - 1]);
#endif
    {   // k_outer_factor on the same buffers (S = D2 + 200 I is diagonally dominant: a valid sweep)
        std::vector<float> hs(h);
        for (int t = 0; t < T; ++t) for (int i = 0; i < n; ++i) { for (int j = 0; j < n; ++j) { float& v = hs[((size_t)t * n + i) * n + j]; v = 0.5f * (v + h[((size_t)t * n + j) * n + i]); } }
        for (int t = 0; t < T; ++t) for (int i = 0; i < n; ++i) hs[((size_t)t * n + i) * n + i] += 200.f;
        float* fo; int32_t* info;
        hipMalloc(&fo, T * 4); hipMalloc(&info, T * 4); hipMemset(info, 0, T * 4);
        OuterArgs oa{a.tv, Ainv, D2, ys, ys, vecs, scal, fo, info, T, 1};
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) { hipMemcpyAsync(D2, hs.data(), hs.size() * 4, hipMemcpyHostToDevice, 0); k_outer_factor<128, 512><<<T, 512>>>(oa); }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
        }
        hipMemcpy(D2, hs.data(), hs.size() * 4, hipMemcpyHostToDevice);
        hipEventRecord(e0);
        k_outer_factor<128, 512><<<T, 512>>>(oa);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("k_outer_factor %.2f us (single launch)\n", ms * 1000);
#if ADKF_STAMP_SMALL
        hipMemcpyFromSymbol(st, HIP_SYMBOL(g_small_stamps), sizeof(st));
        const char* nm[] = {"", "S loads + residual", "sweep", "finish + pivot ratio", "solve", "block_sum", "store S^-1", "C^T e"};
        for (int k = 1; k <= 7; ++k) printf("k_outer_factor %-22s %llu cycles\n", nm[k], st[k] - st[k - 1]);
#endif
    }
    return 0;
}
