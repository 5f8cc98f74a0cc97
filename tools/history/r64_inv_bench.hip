// Timing of the float64 Gauss-Jordan inverse of the ill-conditioned-task path (csrc/refine64.h: r64_inverse_reg, and the in-LDS
// r64_inverse it replaced) alone: 64 workgroups, one 128 x 128 SPD matrix each.  Not part of the library.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I adkf_ift_amd/csrc tools/r64_inv_bench.hip -o tools/r64_inv_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "refine64.h"
using namespace adkf;

__global__ __launch_bounds__(512) void k_inv_reg(double* M, int n, int ld, double* out) {
    __shared__ __attribute__((aligned(16))) double buf[2 * R64_MAXN];
    double logdet;
    const int bad = r64_inverse_reg(M + (size_t)blockIdx.x * ld * ld, n, ld, logdet, buf);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = logdet; out[2 * blockIdx.x + 1] = bad; }
}
__global__ __launch_bounds__(512) void k_inv_lds(double* M, int n, int ld, double* out) {
    __shared__ __attribute__((aligned(16))) double buf[2 * R64_MAXN];
    double logdet;
    const int bad = r64_inverse(M + (size_t)blockIdx.x * ld * ld, n, ld, logdet, buf, buf + R64_MAXN, r64_lds, false);
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = logdet; out[2 * blockIdx.x + 1] = bad; }
}

int main(int argc, char** argv) {
    const int T = 64, n = argc > 1 ? atoi(argv[1]) : 128, ld = 128;
    std::vector<double> h((size_t)T * ld * ld, 0.0);
    unsigned s = 7u;
    for (int t = 0; t < T; ++t)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j <= i; ++j) {
                s = s * 1664525u + 1013904223u;
                const double v = i == j ? 3.0 : 0.02 * (((s >> 8) & 0xffff) / 65536.0 - 0.5);
                h[((size_t)t * ld + i) * ld + j] = v; h[((size_t)t * ld + j) * ld + i] = v;
            }
    double *M, *M0, *out;
    hipMalloc(&M, h.size() * 8); hipMalloc(&M0, h.size() * 8); hipMalloc(&out, T * 16);
    hipMemcpy(M0, h.data(), h.size() * 8, hipMemcpyHostToDevice);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_inv_lds), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 128 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    std::vector<double> res(h.size()), o(2 * T);
    for (int which = 0; which < 2; ++which) {
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipMemcpy(M, M0, h.size() * 8, hipMemcpyDeviceToDevice);
            hipEventRecord(e0);
            if (which == 0) k_inv_reg<<<T, 512>>>(M, n, ld, out); else k_inv_lds<<<T, 512, 128 * 128 * 8>>>(M, n, ld, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = ms < best ? ms : best;
        }
        hipMemcpy(res.data(), M, h.size() * 8, hipMemcpyDeviceToHost);
        hipMemcpy(o.data(), out, T * 16, hipMemcpyDeviceToHost);
        // residual of task 0: max |A Ainv - I|
        double err = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double a = 0.0;
                for (int k = 0; k < n; ++k) a += h[(size_t)i * ld + k] * res[(size_t)k * ld + j];
                err = fmax(err, fabs(a - (i == j ? 1.0 : 0.0)));
            }
        printf("%s  n=%d  %.1f us per launch (64 workgroups)  |A Ainv - I|max %.2e  logdet %.6f  bad %g\n", which == 0 ? "registers" : "LDS      ", n, best * 1000, err, o[0], o[1]);
    }
    return 0;
}
