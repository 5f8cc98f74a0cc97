// Timing of one block step of the blocked sweep (csrc/large.h: k_lg_diag, ProbLgPanel, ProbLgUpdate) at the C5 shape (8 tasks of
// 1024 points), each launch on its own, with the ablation switches of gemm.h (-DADKF_GEMM_ABLATE=bits).  Not part of the library.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I adkf_ift_amd/csrc tools/lg_bench.hip -o tools/lg_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels.h"
#include "large.h"
using namespace adkf;

static inline int grid_for(int T, int tiles) { return ((T + 7) / 8) * 8 * tiles; }

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 8, n = argc > 2 ? atoi(argv[2]) : 1024, step = 3;
    const int tn = n / GT;
    std::vector<float> h((size_t)T * n * n);
    unsigned s = 12345u;
    for (size_t t = 0; t < (size_t)T; ++t)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) {
                s = s * 1664525u + 1013904223u;
                const float v = (i == j) ? 4.f : 0.01f * (((s >> 8) & 0xffff) / 65536.f - 0.5f);
                h[(t * n + i) * n + j] = v; h[(t * n + j) * n + i] = v;
            }
    LgMat m;
    float *M, *Dinv, *C, *F, *logdet, *pext; int32_t* info;
    hipMalloc(&M, h.size() * 4); hipMalloc(&Dinv, (size_t)T * LB * LB * 4); hipMalloc(&C, (size_t)T * LB * n * 4); hipMalloc(&F, (size_t)T * LB * n * 4);
    hipMalloc(&logdet, T * 4); hipMalloc(&pext, T * 8); hipMalloc(&info, T * 4);
    hipMemcpy(M, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipMemset(logdet, 0, T * 4); hipMemset(info, 0, T * 4);
    m.M = M; m.ld = n; m.n_arr = nullptr; m.fit = nullptr; m.Dinv = Dinv; m.Cbuf = C; m.Fbuf = F; m.logdet = logdet; m.pext = pext; m.info = info;
    m.cnt = nullptr; m.T = T; m.vec = true;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    ProbLgPanel pp; pp.m = m; pp.step = step;
    ProbLgUpdate pu; pu.m = m; pu.step = step; pu.tri = 0;
    const int mode = argc > 3 ? atoi(argv[3]) : 1;   // 1: upper-triangle grid (what the library launches), 0: one workgroup per tile of the full square
    const int upd_tiles = (mode & 1) ? tn * (tn + 1) / 2 : tn * tn;
    if (mode & 1) pu.tri = upd_tiles;
    auto grid = [&](int tiles) { return grid_for(T, tiles); };
    const int reps = 50;
    for (int which = 0; which < 3; ++which)
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < reps; ++i) {
                if (which == 0) k_lg_diag<<<grid(1), 512>>>(m, step);
                if (which == 1) k_bgemm<ProbLgPanel><<<grid(2 * tn), 256>>>(pp, T, 2, tn);
                if (which == 2) k_bgemm<ProbLgUpdate><<<grid(upd_tiles), 256>>>(pu, T, tn, tn);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep) printf("mode=%d ablate=%d T=%d n=%d %s  %.2f us per launch\n", mode, (int)ADKF_GEMM_ABLATE, T, n, which == 0 ? "diag  " : which == 1 ? "panel " : "update", ms * 1000 / reps);
        }
    return 0;
}
