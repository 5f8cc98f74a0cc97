// k_dense3 (csrc/dense_x3.h: a dense layer with FP32 products on the BF16 matrix pipe, pipelined) against rocBLAS sgemm on the same
// shape: time per launch, error of both against a host float64 reference on sampled rows.  Not part of the library.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I adkf_ift_amd/csrc tools/x3_gemm_bench.hip -lrocblas -o tools/x3_gemm_bench
//   tools/x3_gemm_bench [M N K]
#include <hip/hip_runtime.h>
#include <rocblas/rocblas.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels.h"
#include "dense_x3.h"
using namespace adkf;

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 65536, N = argc > 2 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 256;
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K);
    unsigned s = 777u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : hA) v = 2.f * rnd();
    for (auto& v : hW) v = 0.25f * rnd();
    float *A, *W, *C0, *C1; unsigned short* Wp;
    hipMalloc(&A, hA.size() * 4); hipMalloc(&W, hW.size() * 4); hipMalloc(&C0, (size_t)M * N * 4); hipMalloc(&C1, (size_t)M * N * 4);
    hipMalloc(&Wp, hW.size() * 2 * 3);
    hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice); hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense3), hipFuncAttributeMaxDynamicSharedMemorySize, D3_LDS_BYTES) != hipSuccess) { printf("LDS opt-in refused\n"); return 1; }
    Dense3Args a{A, K, Wp, (size_t)N * K, nullptr, C1, N, M, N, K};
    const int grid = ((M + D3_TM - 1) / D3_TM) * ((N + D3_TN - 1) / D3_TN);
    rocblas_handle h; rocblas_create_handle(&h);
    const float one = 1.f, zero = 0.f;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop = 2.0 * M * N * K;
    for (int v = 0; v < 2; ++v)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            for (int i = 0; i < 20; ++i) {
                if (v == 0) rocblas_sgemm(h, rocblas_operation_transpose, rocblas_operation_none, N, M, K, &one, W, K, A, K, &zero, C0, N);
                else {
                    k_split3<<<(unsigned)(((size_t)N * K / 2 + 255) / 256), 256>>>(W, Wp, (size_t)N * K / 2, (size_t)N * K);
                    k_dense3<<<grid, D3_NT, D3_LDS_BYTES>>>(a);
                }
            }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%s  %.2f us per call  %.1f TFLOP/s (%s)\n", v ? "k_dense3 (+ weight split)" : "rocBLAS sgemm            ", ms * 1000 / 20, flop / (ms / 20 * 1e-3) / 1e12, hipGetErrorString(hipGetLastError()));
        }
    std::vector<float> o0((size_t)M * N), o1((size_t)M * N);
    hipMemcpy(o0.data(), C0, o0.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(o1.data(), C1, o1.size() * 4, hipMemcpyDeviceToHost);
    double mx[2] = {0, 0}, bias[2] = {0, 0}; size_t cnt = 0;
    for (int i = 0; i < M; i += (M > 512 ? M / 257 : 1))
        for (int j = 0; j < N; ++j) {
            double ref = 0, sc = 0;
            for (int k = 0; k < K; ++k) { const double p = (double)hA[(size_t)i * K + k] * hW[(size_t)j * K + k]; ref += p; sc += fabs(p); }
            const double e0_ = (o0[(size_t)i * N + j] - ref) / sc, e1_ = (o1[(size_t)i * N + j] - ref) / sc;
            if (fabs(e0_) > mx[0]) mx[0] = fabs(e0_);
            if (fabs(e1_) > mx[1]) mx[1] = fabs(e1_);
            bias[0] += e0_; bias[1] += e1_; ++cnt;
        }
    printf("error / sum |a||w|:  rocBLAS max %.3e mean signed %.3e    k_dense3 max %.3e mean signed %.3e\n", mx[0], bias[0] / cnt, mx[1], bias[1] / cnt);
    return 0;
}
