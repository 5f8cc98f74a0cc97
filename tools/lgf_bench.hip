// The blocked sweep of csrc/large.h as three launches per block step against csrc/large_fused.h (update k + sweep k + 1 in one
// launch): the inverses must be BIT-IDENTICAL; times per sweep.  Not part of the library.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I adkf_ift_amd/csrc tools/lgf_bench.hip -o tools/lgf_bench
//   tools/lgf_bench [T] [n] [ld]      (n <= ld: ragged tasks get n, n - 37, n - 74, ... points)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "kernels.h"
#include "large_fused.h"
using namespace adkf;

static inline int grid_for(int T, int tiles) { return ((T + 7) / 8) * 8 * tiles; }
static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

struct Bufs { float *M, *Dinv, *C, *F, *logdet, *pext; int32_t *info, *cnt, *narr; };

static void sweep3(LgMat m, hipStream_t st) {
    const int nb = ceil_div(m.ld, LB), tn = ceil_div(m.ld, GT);
    m.cnt = nullptr;
    for (int step = 0; step < nb; ++step) {
        k_lg_diag<<<grid_for(m.T, 1), 512, 0, st>>>(m, step);
        ProbLgPanel pp; pp.m = m; pp.step = step;
        k_bgemm<ProbLgPanel><<<grid_for(m.T, 2 * tn), 256, 0, st>>>(pp, m.T, 2, tn);
        ProbLgUpdate pu; pu.m = m; pu.step = step; pu.tri = tn * (tn + 1) / 2;
        k_bgemm<ProbLgUpdate><<<grid_for(m.T, pu.tri), 256, 0, st>>>(pu, m.T, tn, tn);
    }
}

static int g_stagger = 0, g_prio = 0;
static int g_dyn = 0;   // extra dynamic LDS per workgroup of the fused kernel: 40000 leaves ONE workgroup per CU (the sweeping workgroup then has its CU to itself)
static void sweepf(LgMat m, float* dinv2, hipStream_t st) {
    const int nb = ceil_div(m.ld, LB), tn = ceil_div(m.ld, GT), npair = lgf_npair(tn);
    float* buf[2] = {m.Dinv, dinv2};
    k_lg_diag<<<grid_for(m.T, 1), 512, 0, st>>>(m, 0);
    for (int step = 0; step < nb; ++step) {
        m.Dinv = buf[step & 1];
        ProbLgPanel pp; pp.m = m; pp.step = step;
        k_bgemm<ProbLgPanel><<<grid_for(m.T, 2 * tn), 256, 0, st>>>(pp, m.T, 2, tn);
        LgStepArgs sa{m, buf[(step + 1) & 1], m.cnt, step, tn, npair, step + 1 < nb ? 1 : 0, g_stagger, g_prio};
        k_lg_update_sweep<<<grid_for(m.T, npair), LGF_NT, g_dyn, st>>>(sa);
    }
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 8, n = argc > 2 ? atoi(argv[2]) : 1024, ld = argc > 3 ? atoi(argv[3]) : n;
    const bool ragged = ld != n;
    g_dyn = argc > 4 ? atoi(argv[4]) : 0;
    g_stagger = argc > 5 ? atoi(argv[5]) : 0;
    g_prio = argc > 6 ? atoi(argv[6]) : 0;
    printf("dynamic LDS %d, stagger %d, sweep priority %d\n", g_dyn, g_stagger, g_prio);
    if (g_dyn > 0 && hipFuncSetAttribute(reinterpret_cast<const void*>(&k_lg_update_sweep), hipFuncAttributeMaxDynamicSharedMemorySize, g_dyn) != hipSuccess) { printf("no LDS opt-in\n"); return 2; }
    std::vector<float> h((size_t)T * ld * ld, 0.f);
    std::vector<int32_t> narr(T);
    unsigned s = 12345u;
    for (int t = 0; t < T; ++t) {
        const int nt = ragged ? (n - 37 * t > 8 ? n - 37 * t : 8) : n;
        narr[t] = nt;
        for (int i = 0; i < nt; ++i)
            for (int j = 0; j <= i; ++j) {
                s = s * 1664525u + 1013904223u;
                const float v = (i == j) ? 4.f + 0.001f * i : 0.01f * (((s >> 8) & 0xffff) / 65536.f - 0.5f);
                h[((size_t)t * ld + i) * ld + j] = v; h[((size_t)t * ld + j) * ld + i] = v;
            }
    }
    Bufs b;
    float* dinv2;
    hipMalloc(&b.M, h.size() * 4); hipMalloc(&b.Dinv, (size_t)T * LB * LB * 4); hipMalloc(&dinv2, (size_t)T * LB * LB * 4);
    hipMalloc(&b.C, (size_t)T * LB * ld * 4); hipMalloc(&b.F, (size_t)T * LB * ld * 4);
    hipMalloc(&b.logdet, T * 4); hipMalloc(&b.pext, T * 8); hipMalloc(&b.info, T * 4); hipMalloc(&b.cnt, T * 4); hipMalloc(&b.narr, T * 4);
    hipMemcpy(b.narr, narr.data(), T * 4, hipMemcpyHostToDevice);
    hipMemset(b.cnt, 0xff, T * 4);   // garbage: the sweep's first launch has to zero it
    LgMat m;
    m.M = b.M; m.ld = ld; m.n_arr = ragged ? b.narr : nullptr; m.fit = nullptr; m.Dinv = b.Dinv; m.Cbuf = b.C; m.Fbuf = b.F;
    m.logdet = b.logdet; m.pext = b.pext; m.info = b.info; m.cnt = b.cnt; m.T = T; m.vec = (ld & 3) == 0;
    std::vector<float> r3(h.size()), rf(h.size());
    std::vector<float> l3(T), lf(T), p3(2 * T), pf(2 * T);
    std::vector<int32_t> i3(T), if_(T);
    hipMemcpy(b.M, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    sweep3(m, 0);
    hipDeviceSynchronize();
    printf("three launches: %s\n", hipGetErrorString(hipGetLastError()));
    hipMemcpy(r3.data(), b.M, h.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(l3.data(), b.logdet, T * 4, hipMemcpyDeviceToHost); hipMemcpy(p3.data(), b.pext, T * 8, hipMemcpyDeviceToHost); hipMemcpy(i3.data(), b.info, T * 4, hipMemcpyDeviceToHost);
    int bad_total = 0;
    for (int rep = 0; rep < 5; ++rep) {   // several runs: a stale read would come and go
        hipMemcpy(b.M, h.data(), h.size() * 4, hipMemcpyHostToDevice);
        sweepf(m, dinv2, 0);
        hipDeviceSynchronize();
        hipMemcpy(rf.data(), b.M, h.size() * 4, hipMemcpyDeviceToHost);
        hipMemcpy(lf.data(), b.logdet, T * 4, hipMemcpyDeviceToHost); hipMemcpy(pf.data(), b.pext, T * 8, hipMemcpyDeviceToHost); hipMemcpy(if_.data(), b.info, T * 4, hipMemcpyDeviceToHost);
        size_t bad = 0;
        for (int t = 0; t < T; ++t)
            for (int i = 0; i < narr[t]; ++i)
                for (int j = 0; j < narr[t]; ++j) {
                    const size_t q = ((size_t)t * ld + i) * ld + j;
                    if (memcmp(&r3[q], &rf[q], 4) != 0) { if (bad < 5) printf("  differs t=%d (%d,%d): %g vs %g\n", t, i, j, r3[q], rf[q]); ++bad; }
                }
        const bool scal = memcmp(l3.data(), lf.data(), T * 4) == 0 && memcmp(p3.data(), pf.data(), T * 8) == 0 && memcmp(i3.data(), if_.data(), T * 4) == 0;
        printf("fused run %d: %s, %zu elements differ, log-determinants / pivot extremes / info %s (logdet[0] = %g, M[0][0] = %g)\n", rep, hipGetErrorString(hipGetLastError()), bad, scal ? "equal" : "DIFFER", lf[0], rf[0]);
        bad_total += bad != 0 || !scal;
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 20;
    for (int which = 0; which < 2; ++which)
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < reps; ++i) { if (which == 0) sweep3(m, 0); else sweepf(m, dinv2, 0); }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("T=%d n=%d ld=%d %s  %.1f us per sweep (%d block steps: %.1f us each)\n", T, n, ld, which == 0 ? "three launches" : "fused         ", ms * 1000 / reps,
                            ceil_div(ld, LB), ms * 1000 / reps / ceil_div(ld, LB));
        }
    // the launches of one block step on their own (step 3 of 8: the state of M does not matter for the time)
    {
        const int tn = ceil_div(ld, GT), npair = lgf_npair(tn), step = ceil_div(ld, LB) > 3 ? 3 : 0;
        ProbLgPanel pp; pp.m = m; pp.step = step;
        ProbLgUpdate pu; pu.m = m; pu.step = step; pu.tri = tn * (tn + 1) / 2;
        const char* nm[5] = {"k_lg_diag", "panel", "update (three-launch path)", "update + sweep (fused)", "update alone in the fused kernel (look = 0)"};
        for (int which = 0; which < 5; ++which)
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                for (int i = 0; i < 50; ++i) {
                    if (which == 0) k_lg_diag<<<grid_for(T, 1), 512>>>(m, step);
                    if (which == 1) k_bgemm<ProbLgPanel><<<grid_for(T, 2 * tn), 256>>>(pp, T, 2, tn);
                    if (which == 2) k_bgemm<ProbLgUpdate><<<grid_for(T, pu.tri), 256>>>(pu, T, tn, tn);
                    if (which >= 3) { LgStepArgs sa{m, dinv2, m.cnt, step, tn, npair, which == 3 ? 1 : 0, g_stagger, g_prio}; k_lg_update_sweep<<<grid_for(T, npair), LGF_NT, g_dyn>>>(sa); }
                }
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (rep) printf("   %-46s %.2f us per launch\n", nm[which], ms * 1000 / 50);
            }
    }
    return bad_total ? 1 : 0;
}
