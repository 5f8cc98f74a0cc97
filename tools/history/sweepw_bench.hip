// Correctness + timing of the wave-owned sweep (csrc/factor_w.h) against the blocked one (csrc/factor.h) and a host
// float64 inverse.  Not part of the library.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I adkf_ift_amd/csrc tools/sweepw_bench.hip -o tools/sweepw_bench
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
#include "factor.h"
using namespace adkf;

template <class SW, class SM>
__global__ __launch_bounds__(512) void k_sweep(const float* A, float* Ainv, float* aux, const float* y, int n, int reps) {
    constexpr int RB = SW::RB, CB = SW::CB;
    __shared__ SM sm;
    const float* At = A + (size_t)blockIdx.x * 128 * 128;
    float a[RB][CB], m[RB][CB];
    for (int r = 0; r < RB; ++r) for (int c = 0; c < CB; ++c) {
        const int i = SW::row(r), j = SW::col(c);
        a[r][c] = (i < n && j < n) ? At[i * 128 + j] : (i == j ? 1.f : 0.f);
    }
    if (threadIdx.x < 128) sm.vec_in[threadIdx.x] = threadIdx.x < n ? y[threadIdx.x] : 0.f;
    unsigned long long t_begin = 0;
    if (threadIdx.x == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_begin) :: "memory");
    for (int it = 0; it < reps; ++it) {
        for (int r = 0; r < RB; ++r) for (int c = 0; c < CB; ++c) m[r][c] = a[r][c];
        __syncthreads();
        SW::run(m, n, sm);
    }
    unsigned long long t_end = 0;
    if (threadIdx.x == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end) :: "memory");
    float logdet;
    const int info = SW::finish(n, sm, logdet);
    SW::solve(m, sm.vec_in, sm.vec_out);
    float* Ao = Ainv + (size_t)blockIdx.x * 128 * 128;
    for (int r = 0; r < RB; ++r) for (int c = 0; c < CB; ++c) Ao[SW::row(r) * 128 + SW::col(c)] = -m[r][c];
    float* ax = aux + (size_t)blockIdx.x * 256;
    if (threadIdx.x < 128) ax[threadIdx.x] = sm.vec_out[threadIdx.x];
#if ADKF_STAMP
    if (blockIdx.x == 3 && threadIdx.x < 128) ((unsigned long long*)(aux + (size_t)gridDim.x * 256))[threadIdx.x] = sm.stamp[threadIdx.x];
#endif
    if (threadIdx.x == 0) { ax[128] = logdet; ax[129] = (float)info; ((unsigned long long*)(ax + 130))[0] = (t_end - t_begin) / reps; }
}

static void host_inverse(const std::vector<double>& A, int n, std::vector<double>& X, double& logdet) {
    std::vector<double> M(A);
    X.assign((size_t)n * n, 0.0);
    for (int i = 0; i < n; ++i) X[(size_t)i * n + i] = 1.0;
    logdet = 0.0;
    for (int k = 0; k < n; ++k) {   // Gauss-Jordan in float64 (SPD: no pivoting needed)
        const double p = M[(size_t)k * n + k];
        logdet += std::log(p);
        for (int j = 0; j < n; ++j) { M[(size_t)k * n + j] /= p; X[(size_t)k * n + j] /= p; }
        for (int i = 0; i < n; ++i) if (i != k) {
            const double f = M[(size_t)i * n + k];
            if (f == 0.0) continue;
            for (int j = 0; j < n; ++j) { M[(size_t)i * n + j] -= f * M[(size_t)k * n + j]; X[(size_t)i * n + j] -= f * X[(size_t)k * n + j]; }
        }
    }
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 256, reps = 20;
    std::vector<float> A((size_t)T * 128 * 128), y(128);
    srand(1);
    for (int i = 0; i < 128; ++i) y[i] = (float)rand() / RAND_MAX - 0.5f;
    for (int t = 0; t < T; ++t) {   // RBF kernel matrix of random 3-d points + noise: a GP-like SPD matrix, cond ~ 1e2
        float pts[128][3];
        for (int i = 0; i < 128; ++i) for (int k = 0; k < 3; ++k) pts[i][k] = 3.f * (float)rand() / RAND_MAX;
        for (int i = 0; i < 128; ++i) for (int j = 0; j < 128; ++j) {
            float d2 = 0.f; for (int k = 0; k < 3; ++k) d2 += (pts[i][k] - pts[j][k]) * (pts[i][k] - pts[j][k]);
            A[((size_t)t * 128 + i) * 128 + j] = 0.7f * expf(-0.5f * d2) + (i == j ? 0.1f : 0.f);
        }
    }
    float *dA, *dInv, *daux, *dy;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dInv, A.size() * 4); hipMalloc(&daux, (size_t)T * 256 * 4 + 1024); hipMalloc(&dy, 128 * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dy, y.data(), 128 * 4, hipMemcpyHostToDevice);
    std::vector<float> inv(A.size()), aux((size_t)T * 256);
    int rc = 0;
    for (int variant = 0; variant < 2; ++variant) {
        for (int n : {128, 127, 100, 70, 17, 5, 1}) {
            hipMemset(dInv, 0, A.size() * 4);
            if (variant == 0) k_sweep<SweepBlk<128, 512>, SweepSmemBlk<128, 512>><<<T, 512>>>(dA, dInv, daux, dy, n, 1);
            else k_sweep<Sweep<128, 512>, SweepSmem<128, 512>><<<T, 512>>>(dA, dInv, daux, dy, n, 1);
            if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 2; }
            hipMemcpy(inv.data(), dInv, A.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(aux.data(), daux, aux.size() * 4, hipMemcpyDeviceToHost);
            double worst = 0, worst_alpha = 0, worst_ld = 0;
            for (int t : {0, 1, T / 2, T - 1}) {
                std::vector<double> Ad((size_t)n * n), X; double ld;
                for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Ad[(size_t)i * n + j] = A[((size_t)t * 128 + i) * 128 + j];
                host_inverse(Ad, n, X, ld);
                double mx = 0, err = 0;
                for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { mx = fmax(mx, fabs(X[(size_t)i * n + j])); err = fmax(err, fabs(X[(size_t)i * n + j] - inv[((size_t)t * 128 + i) * 128 + j])); }
                worst = fmax(worst, err / mx);
                double amx = 0, aerr = 0;
                for (int i = 0; i < n; ++i) { double s = 0; for (int j = 0; j < n; ++j) s += X[(size_t)i * n + j] * y[j]; amx = fmax(amx, fabs(s)); aerr = fmax(aerr, fabs(s - aux[(size_t)t * 256 + i])); }
                worst_alpha = fmax(worst_alpha, aerr / amx);
                worst_ld = fmax(worst_ld, fabs(ld - aux[(size_t)t * 256 + 128]) / fmax(1.0, fabs(ld)));
                if (aux[(size_t)t * 256 + 129] != 0.f) { printf("info != 0\n"); rc = 1; }
            }
            printf("%s n=%3d: inverse rel err %.2e, A^-1 y rel err %.2e, logdet rel err %.2e\n", variant ? "wave-owned" : "blocked   ", n, worst, worst_alpha, worst_ld);
            if (!(worst < 2e-4 && worst_alpha < 2e-4 && worst_ld < 1e-5)) rc = 1;
        }
    }
#if ADKF_STAMP
    {
        k_sweep<Sweep<128, 512>, SweepSmem<128, 512>><<<T, 512>>>(dA, dInv, daux, dy, 128, 1);
        hipDeviceSynchronize();
        std::vector<unsigned long long> st(128);
        hipMemcpy(st.data(), daux + (size_t)T * 256, 1024, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull;
        for (int w = 0; w < 8; ++w) if (st[w * 16] && st[w * 16] < t0) t0 = st[w * 16];
        printf("stamps of step %d (0 after barrier | chain: 1 prio, 2 row updated, 3 bpermute issued, 4 C stored, 5 GJ half, 6 GJ done, 7 F stored | 8 pending done, 9 bulk done)\n", ADKF_STAMP);
        for (int w = 0; w < 8; ++w) { printf("wave %d:", w); for (int k = 0; k < 10; ++k) printf(" %6lld", st[w * 16 + k] >= t0 && st[w*16+k] - t0 < 100000 ? (long long)(st[w * 16 + k] - t0) : -1ll); printf("\n"); }
    }
#endif
    // timing, n = 128
    for (int variant = 0; variant < 2; ++variant) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        float best = 1e9f;
        for (int rep = 0; rep < 5; ++rep) {
            hipEventRecord(e0);
            if (variant == 0) k_sweep<SweepBlk<128, 512>, SweepSmemBlk<128, 512>><<<T, 512>>>(dA, dInv, daux, dy, 128, reps);
            else k_sweep<Sweep<128, 512>, SweepSmem<128, 512>><<<T, 512>>>(dA, dInv, daux, dy, 128, reps);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            best = fminf(best, ms * 1e3f / reps);
        }
        hipMemcpy(aux.data(), daux, aux.size() * 4, hipMemcpyDeviceToHost);
        printf("%s: %.1f us per sweep (T=%d), %llu ticks per sweep in workgroup 8\n", variant ? "wave-owned" : "blocked   ", best, T,
               ((unsigned long long*)(aux.data() + 8 * 256 + 130))[0]);
    }
    printf(rc ? "FAILED\n" : "OK\n");
    return rc;
}
