// Timing of the staged FP64 product of the float64 path (csrc/refine64.h: r64_mm_staged) alone: 64 workgroups, one 128^3 product each,
// operands in global memory as in the path.  Not part of the library.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I adkf_ift_amd/csrc tools/r64_mm_bench.hip -o tools/r64_mm_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "refine64.h"
using namespace adkf;

__global__ __launch_bounds__(512) void k_mm(const double* A, const double* B, double* C, int n, int ld, int reps) {
    const double* a = A + (size_t)blockIdx.x * ld * ld;
    const double* b = B + (size_t)blockIdx.x * ld * ld;
    double* c = C + (size_t)blockIdx.x * ld * ld;
    for (int r = 0; r < reps; ++r)
        r64_mm(n, n, n, [=](int i, int k) { return a[(size_t)i * ld + k]; }, [=](int k, int j) { return b[(size_t)k * ld + j]; },
               [=](int i, int j, double v) { c[(size_t)i * ld + j] = v; }, r64_lds);
}

int main(int argc, char** argv) {
    const int T = 64, n = argc > 1 ? atoi(argv[1]) : 128, ld = 128, reps = 8;
    std::vector<double> h((size_t)T * ld * ld);
    unsigned s = 7u;
    for (auto& v : h) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.0 - 0.5; }
    double *A, *B, *C;
    hipMalloc(&A, h.size() * 8); hipMalloc(&B, h.size() * 8); hipMalloc(&C, h.size() * 8);
    hipMemcpy(A, h.data(), h.size() * 8, hipMemcpyHostToDevice); hipMemcpy(B, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_mm), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 128 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
        hipEventRecord(e0);
        k_mm<<<T, 512, 128 * 128 * 8>>>(A, B, C, n, ld, reps);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms < best ? ms : best;
    }
    std::vector<double> c(h.size());
    hipMemcpy(c.data(), C, h.size() * 8, hipMemcpyDeviceToHost);
    double want = 0.0;
    for (int k = 0; k < n; ++k) want += h[(size_t)3 * ld + k] * h[(size_t)k * ld + 5];
    printf("r64_mm staged  n=%d  %.1f us per product (64 workgroups, %d products per launch)  C[3][5] %.12f (host %.12f)\n", n, best * 1000 / reps, reps, c[(size_t)3 * ld + 5], want);
    return 0;
}
