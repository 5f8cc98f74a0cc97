// The streaming forms of csrc/dense_x3.h against the forms they replace, at the C2 stand-in feature map's shape by default:
//   forward   k_dense3 (one chunk of loads in flight)  vs  k_dense3_sk<K / 32> (persistent, a row tile's whole K extent in registers)
//   gradient  k_dense3_tn                              vs  k_dense3_tnd<2 | 3 | 4> (deeper prefetch, XCD-aware order)
// Time per launch (HIP events, 20 launches, third repetition), and a bit-for-bit comparison of the results (same products, same order).
// Not part of the library.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I adkf_ift_amd/csrc tools/x3_stream_bench.hip -o tools/x3_stream_bench
//   tools/x3_stream_bench [M N K]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "kernels.h"
#include "dense_x3.h"
using namespace adkf;

#define CK(x_) do { hipError_t e_ = (x_); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

template <class F> static float time_us(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 0.f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) f();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        best = ms * 1000.f / 20.f;
    }
    return best;
}

template <class K> static bool optin(K k) { return hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, D3_LDS_BYTES) == hipSuccess; }

int main(int argc, char** argv) {
    const int M = argc > 1 ? atoi(argv[1]) : 65536, N = argc > 2 ? atoi(argv[2]) : 256, K = argc > 3 ? atoi(argv[3]) : 256;
    if (K % 32 || K < 64 || K > 256 || (K / 32 != 2 && K / 32 != 4 && K / 32 != 8)) { printf("K must be 64, 128 or 256\n"); return 1; }
    std::vector<float> hA((size_t)M * K), hW((size_t)N * K), hG((size_t)M * N);
    unsigned s = 777u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    for (auto& v : hA) v = 2.f * rnd();
    for (auto& v : hW) v = 0.25f * rnd();
    for (auto& v : hG) v = rnd();
    float *A, *W, *G, *C0, *C1; unsigned short* Wp;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&W, hW.size() * 4)); CK(hipMalloc(&G, hG.size() * 4));
    CK(hipMalloc(&C0, (size_t)M * N * 4)); CK(hipMalloc(&C1, (size_t)M * N * 4)); CK(hipMalloc(&Wp, hW.size() * 2 * 3));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W, hW.data(), hW.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(G, hG.data(), hG.size() * 4, hipMemcpyHostToDevice));
    if (!optin(&k_dense3) || !optin(&k_dense3_sk<2>) || !optin(&k_dense3_sk<4>) || !optin(&k_dense3_sk<8>) || !optin(&k_dense3_tn) ||
        !optin(&k_dense3_tnd<2>) || !optin(&k_dense3_tnd<3>) || !optin(&k_dense3_tnd<4>)) { printf("LDS opt-in refused\n"); return 1; }
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    const int cus = prop.multiProcessorCount;
    k_split3<<<(unsigned)(((size_t)N * K / 2 + 255) / 256), 256>>>(W, Wp, (size_t)N * K / 2, (size_t)N * K);
    const double flop = 2.0 * M * N * K;

    // ---- forward ----
    const int tiles_m = (M + D3_TM - 1) / D3_TM, tiles_n = (N + D3_TN - 1) / D3_TN;
    Dense3Args a0{A, K, Wp, (size_t)N * K, nullptr, C0, N, M, N, K}, a1 = a0;
    a1.C = C1;
    CK(hipMemset(C0, 0xff, (size_t)M * N * 4)); CK(hipMemset(C1, 0x7f, (size_t)M * N * 4));
    const float t_old = time_us([&]() { k_dense3<<<tiles_m * tiles_n, D3_NT, D3_LDS_BYTES>>>(a0); });
    printf("forward  k_dense3          %8.2f us  %6.1f TFLOP/s (%s)\n", t_old, flop / (t_old * 1e-6) / 1e12, hipGetErrorString(hipGetLastError()));
    for (int mult = 1; mult <= 2; ++mult) {
        const int grid = tiles_m < cus * mult ? tiles_m : cus * mult;
        auto launch = [&]() {
            if (K == 256) k_dense3_sk<8><<<grid, D3_NT, D3_LDS_BYTES>>>(a1);
            else if (K == 128) k_dense3_sk<4><<<grid, D3_NT, D3_LDS_BYTES>>>(a1);
            else k_dense3_sk<2><<<grid, D3_NT, D3_LDS_BYTES>>>(a1);
        };
        const float t_new = time_us(launch);
        CK(hipDeviceSynchronize());
        std::vector<float> o0((size_t)M * N), o1((size_t)M * N);
        CK(hipMemcpy(o0.data(), C0, o0.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(o1.data(), C1, o1.size() * 4, hipMemcpyDeviceToHost));
        const bool same = memcmp(o0.data(), o1.data(), o0.size() * 4) == 0;
        printf("forward  k_dense3_sk grid %4d %8.2f us  %6.1f TFLOP/s  bit-identical to k_dense3: %s (%s)\n", grid, t_new, flop / (t_new * 1e-6) / 1e12,
               same ? "yes" : "NO", hipGetErrorString(hipGetLastError()));
        if (mult == 1) {   // against float64 on sampled rows
            double mx = 0;
            for (int i = 0; i < M; i += (M > 512 ? M / 257 : 1))
                for (int j = 0; j < N; ++j) {
                    double ref = 0, sc = 0;
                    for (int k = 0; k < K; ++k) { const double p_ = (double)hA[(size_t)i * K + k] * hW[(size_t)j * K + k]; ref += p_; sc += fabs(p_); }
                    const double e = fabs(o1[(size_t)i * N + j] - ref) / sc;
                    if (e > mx) mx = e;
                }
            printf("         k_dense3_sk error / sum |a||w| against float64: %.3e\n", mx);
        }
        CK(hipMemset(C1, 0x7f, (size_t)M * N * 4));
    }

    // ---- weight gradient dW[N, K] = G[M, N]^T A[M, K] ----
    const int tiles = ((N + D3_TM - 1) / D3_TM) * ((K + D3_TN - 1) / D3_TN);
    long long sp = (4LL * cus + tiles - 1) / tiles;
    const long long max_s = (M + 4 * GK - 1) / (4 * GK);
    if (sp > max_s) sp = max_s;
    if (sp > 64) sp = 64;
    if (sp < 1) sp = 1;
    int rps = (int)((M + sp - 1) / sp);
    rps = (rps + GK - 1) / GK * GK;
    const int splits = (M + rps - 1) / rps;
    float *P0, *P1;
    const size_t pn = (size_t)splits * N * K;
    CK(hipMalloc(&P0, pn * 4)); CK(hipMalloc(&P1, pn * 4));
    Dense3TnArgs t0{G, N, A, K, P0, M, N, K, rps}, t1 = t0;
    t1.part = P1;
    const float tt_old = time_us([&]() { k_dense3_tn<<<dim3(tiles, splits), D3_NT, D3_LDS_BYTES>>>(t0); });
    printf("gradient k_dense3_tn       %8.2f us  %6.1f TFLOP/s  (%d tiles x %d row ranges of %d) (%s)\n", tt_old, flop / (tt_old * 1e-6) / 1e12, tiles, splits, rps,
           hipGetErrorString(hipGetLastError()));
    std::vector<float> q0(pn), q1(pn);
    CK(hipMemcpy(q0.data(), P0, pn * 4, hipMemcpyDeviceToHost));
    for (int depth = 2; depth <= 4; ++depth) {
        CK(hipMemset(P1, 0x7f, pn * 4));
        auto launch = [&]() {
            if (depth == 2) k_dense3_tnd<2><<<tiles * splits, D3_NT, D3_LDS_BYTES>>>(t1, tiles, splits);
            else if (depth == 3) k_dense3_tnd<3><<<tiles * splits, D3_NT, D3_LDS_BYTES>>>(t1, tiles, splits);
            else k_dense3_tnd<4><<<tiles * splits, D3_NT, D3_LDS_BYTES>>>(t1, tiles, splits);
        };
        const float tt = time_us(launch);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(q1.data(), P1, pn * 4, hipMemcpyDeviceToHost));
        printf("gradient k_dense3_tnd<%d>   %8.2f us  %6.1f TFLOP/s  bit-identical to k_dense3_tn: %s (%s)\n", depth, tt, flop / (tt * 1e-6) / 1e12,
               memcmp(q0.data(), q1.data(), pn * 4) == 0 ? "yes" : "NO", hipGetErrorString(hipGetLastError()));
    }
    return 0;
}
