// Timing of the batched distance GEMM (csrc/gemm.h + ProbDistMulti) at the C2 shape, with the ablation switches of gemm.h
// (-DADKF_GEMM_ABLATE=bits).  Not part of the library.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I adkf_ift_amd/csrc tools/gemm_bench.hip -o tools/gemm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "kernels.h"
using namespace adkf;

int main() {
    const int T = 256, n = 128, d = 256;
    std::vector<float> h((size_t)T * n * d);
    unsigned s = 12345u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
    float *Zs, *Zq, *mean, *nrm, *Dss, *Dqs, *Dqq;
    hipMalloc(&Zs, h.size() * 4); hipMalloc(&Zq, h.size() * 4); hipMalloc(&mean, (size_t)T * d * 4); hipMalloc(&nrm, (size_t)T * n * 4);
    hipMalloc(&Dss, (size_t)T * n * n * 4); hipMalloc(&Dqs, (size_t)T * n * n * 4); hipMalloc(&Dqq, (size_t)T * n * n * 4);
    hipMemcpy(Zs, h.data(), h.size() * 4, hipMemcpyHostToDevice); hipMemcpy(Zq, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(mean, 0, (size_t)T * d * 4); hipMemset(nrm, 0, (size_t)T * n * 4);
    ProbDist p;
    p.mean = mean; p.d = d; p.vec = true; p.n_x = nullptr; p.n_y = nullptr; p.x_ld = n; p.y_ld = n;
    ProbDistMulti pm;
    pm.vec = true;
    p.X = Zs; p.Y = Zs; p.symmetric = true; p.D2 = Dss; pm.s0 = p;
    p.X = Zq; p.Y = Zs; p.symmetric = false; p.D2 = Dqs; pm.s1 = p;
    p.X = Zq; p.Y = Zq; p.symmetric = true; p.D2 = Dqq; pm.s2 = p;
    pm.tn0 = pm.tn1 = pm.tn2 = 2; pm.end0 = 4; pm.end1 = 8;
    const int total = 12;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) k_bgemm<ProbDistMulti, GT><<<T * total, 256>>>(pm, T, 1, total);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double flop = 2.0 * T * 10 * 64 * 64 * d;
        printf("ablate=%d  %.2f us per launch  %.1f TFLOP/s\n", (int)ADKF_GEMM_ABLATE, ms * 1000 / 20, flop / (ms / 20 * 1e-3) / 1e12);
    }
    std::vector<float> out(16);
    hipMemcpy(out.data(), Dqs, 64, hipMemcpyDeviceToHost);
    printf("D2qs[0..3] = %g %g %g %g\n", out[0], out[1], out[2], out[3]);
    return 0;
}
