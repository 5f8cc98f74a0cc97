// Phase stamps and launch time of k_hyper (csrc/hyper.h) at the C2 shape (256 tasks x 128 + 128 points).  Not part of the library.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -DADKF_STAMP_SMALL=1 -I adkf_ift_amd/csrc -I tools tools/hyper_bench.hip -o tools/hyper_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "variants/dz.h"
using namespace adkf;

int main() {
    const int T = 256, n = 128;
    const size_t NN = (size_t)T * n * n;
    std::vector<float> d2(NN), ai(NN);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f; };
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) {
                const float v = i == j ? 0.f : 200.f + 100.f * rnd();          // squared distances of the C2 order (d = 256)
                d2[((size_t)t * n + i) * n + j] = v; d2[((size_t)t * n + j) * n + i] = v;
                const float a = i == j ? 5.f + rnd() : 0.02f * (rnd() - 0.5f);  // a diagonally dominant "A^-1"
                ai[((size_t)t * n + i) * n + j] = a; ai[((size_t)t * n + j) * n + i] = a;
            }
    float *st1, *st2, *st3, *Ainv, *Dss, *Dqs, *Dqq, *ys, *yq, *pri, *scal, *vecs, *Wss, *Wqs, *Wqq, *fo, *gp, *vo, *Ho;
    int32_t* info;
    for (float** p : {&Ainv, &Dss, &Dqs, &Dqq, &Wss, &Wqs, &Wqq, &st1, &st2, &st3}) hipMalloc(p, NN * 4);
    hipMalloc(&ys, (size_t)T * n * 4); hipMalloc(&yq, (size_t)T * n * 4); hipMalloc(&pri, T * 16); hipMalloc(&scal, (size_t)T * NSCAL * 4);
    hipMalloc(&vecs, (size_t)T * NVEC * n * 4); hipMalloc(&fo, T * 4); hipMalloc(&gp, T * 12); hipMalloc(&vo, T * 12); hipMalloc(&Ho, T * 36); hipMalloc(&info, T * 4);
    hipMemcpy(Ainv, ai.data(), NN * 4, hipMemcpyHostToDevice);
    for (float* p : {Dss, Dqs, Dqq}) hipMemcpy(p, d2.data(), NN * 4, hipMemcpyHostToDevice);
    std::vector<float> sc((size_t)T * NSCAL, 0.5f), y((size_t)T * n), al((size_t)T * NVEC * n, 0.f);
    for (int t = 0; t < T; ++t) { float* q = &sc[(size_t)t * NSCAL]; q[S_NOISE] = 0.1f; q[S_OS] = 0.7f; q[S_LS] = 16.f; q[S_CONDA] = 1.5f; q[S_AREF] = 0.f; }
    for (auto& v : y) v = rnd() > 0.5f ? 1.f : -1.f;
    for (int t = 0; t < T; ++t) for (int i = 0; i < n; ++i) al[((size_t)t * NVEC + V_ALPHA) * n + i] = 0.2f * (rnd() - 0.5f);
    hipMemcpy(scal, sc.data(), sc.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(ys, y.data(), y.size() * 4, hipMemcpyHostToDevice); hipMemcpy(yq, y.data(), y.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(vecs, al.data(), al.size() * 4, hipMemcpyHostToDevice);
    std::vector<float> pr(T * 4, 0.25f);
    hipMemcpy(pri, pr.data(), pr.size() * 4, hipMemcpyHostToDevice);
    HyperArgs a{};
    a.tv.n_s = nullptr; a.tv.n_q = nullptr; a.tv.ns_ld = n; a.tv.nq_ld = n; a.tv.vld = n; a.tv.kind = 0; a.tv.scal = scal; a.tv.vecs = vecs; a.tv.vec = true;
    a.Ainv = Ainv; a.D2ss = Dss; a.D2qs = Dqs; a.D2qq = Dqq; a.y_s = ys; a.y_q = yq; a.priors = pri; a.Wss = Wss; a.Wqs = Wqs; a.Wqq = Wqq; a.stash_ss = st1; a.stash_qs = st2; a.stash_qq = st3;
    a.vecs = vecs; a.scal = scal; a.f_out = fo; a.info = info; a.g_phi_out = gp; a.v_out = vo; a.H_out = Ho;
    a.T = T; a.reset_info = 1; a.with_hessian = 1; a.flags = 0; a.dirscale = 1.f; a.corrscale = 1.f; a.refine_thresh = 3.f;
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_hyper<true, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)HY_LDS_BYTES) != hipSuccess) { printf("no LDS opt-in\n"); return 1; }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) k_hyper<true, 0><<<T, HY_NT, HY_LDS_BYTES>>>(a);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("k_hyper %.2f us per launch (%s)\n", ms * 1000 / 20, hipGetErrorString(hipGetLastError()));
    }
    float f0[4]; hipMemcpy(f0, fo, 16, hipMemcpyDeviceToHost);
    printf("f_out[0..3] = %g %g %g %g\n", f0[0], f0[1], f0[2], f0[3]);
#if ADKF_STAMP_SMALL
    unsigned long long st[32];
    hipMemcpyFromSymbol(st, HIP_SYMBOL(g_small_stamps), sizeof(st));
    const char* nm[] = {"", "loads A^-1, kappa_qs", "G1 C^T", "store C, r (+refine)", "G2 C K_sq", "kqs regs, S epilogue", "sweep", "finish pivr solve q",
                        "W_qq + store S^-1", "colsum C^T e", "G3 S^-1 C", "OC epilogue", "store OC^T, G4 M_A", "M_A epilogue", "reductions g_out",
                        "put A^-1, store G", "colsums b g d", "G5 P", "traces 1", "store P^T, trPP", "sum, H, v, w", "B' in place", "G6 mixed",
                        "mixed epilogue", "store W_ss"};
    unsigned long long tot = 0;
    for (int k = 1; k <= 24; ++k) { printf("%-22s %8llu cycles\n", nm[k], st[k] - st[k - 1]); tot += st[k] - st[k - 1]; }
    printf("total %llu cycles (s_memtime ticks at 100 MHz x ... see DESIGN)\n", tot);
#endif
    {   // k_dz on the same weight matrices
        const int d = 256;
        float *Zs, *Zq, *dZs, *dZq;
        for (float** p : {&Zs, &Zq, &dZs, &dZq}) hipMalloc(p, (size_t)T * n * d * 4);
        std::vector<float> z((size_t)T * n * d);
        for (auto& v : z) v = rnd() - 0.5f;
        hipMemcpy(Zs, z.data(), z.size() * 4, hipMemcpyHostToDevice); hipMemcpy(Zq, z.data(), z.size() * 4, hipMemcpyHostToDevice);
        hipMemset(Wss, 0, NN * 4); hipMemset(Wqs, 0, NN * 4); hipMemset(Wqq, 0, NN * 4);
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dz), hipFuncAttributeMaxDynamicSharedMemorySize, (int)DZ_LDS_BYTES) != hipSuccess) { printf("no LDS opt-in\n"); return 1; }
        DzArgs da{Wss, Wqs, Wqq, Zs, Zq, dZs, dZq, d, T};
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) k_dz<<<T, HY_NT, DZ_LDS_BYTES>>>(da);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            printf("k_dz %.2f us per launch (%s)\n", ms * 1000 / 20, hipGetErrorString(hipGetLastError()));
        }
#if ADKF_STAMP_SMALL
        unsigned long long st2[32];
        hipMemcpyFromSymbol(st2, HIP_SYMBOL(g_small_stamps), sizeof(st2));
        for (int k = 1; k < 20; ++k) printf("k_dz stamp %2d -> %2d: %8llu cycles%s\n", k - 1, k, st2[k] - st2[k - 1], (k & 1) ? "   <- product / output" : "");
#endif
    }
    return 0;
}
