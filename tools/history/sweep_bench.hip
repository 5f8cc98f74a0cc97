// Micro-benchmark / ablation harness for the register-resident sweep (factor.h).  Not part of the library.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I adkf_ift_amd/csrc tools/sweep_bench.hip -o /tmp/sweep_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "factor.h"
using namespace adkf;

template <int NMAX, int NT, int VAR>
__global__ __launch_bounds__(NT) void k_sweep(const float* A, float* out, int n, int reps) {
    using SW = SweepBlk<NMAX, NT>;
    constexpr int RB = SW::RB, CB = SW::CB;
    __shared__ SweepSmemBlk<NMAX, NT> sm;
    const int j0 = SW::bc() * CB;
    const float* At = A + (size_t)blockIdx.x * n * n;
    float a[RB][CB], m[RB][CB];
    for (int r = 0; r < RB; ++r) for (int c = 0; c < CB; ++c) a[r][c] = At[SW::row(r) * n + j0 + c];
    float acc = 0.f;
    unsigned long long t_begin = 0;
    if (threadIdx.x == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_begin) :: "memory");
    for (int it = 0; it < reps; ++it) {
        for (int r = 0; r < RB; ++r) for (int c = 0; c < CB; ++c) m[r][c] = a[r][c];
        __syncthreads();
        if (VAR == 0) SW::run(m, n, sm);
        if (VAR == 1) {  // update only: no publish (stale vectors), same barriers
            const int nq = n / SW::B;
            for (int q = 0; q < nq; ++q) { __syncthreads(); SW::step(m, q, sm); }
        }
        if (VAR == 3) {  // barriers only
            const int nq = n / SW::B;
            for (int q = 0; q < nq; ++q) { __syncthreads(); asm volatile("" ::: "memory"); }
        }
        acc += m[0][0];
    }
    if (threadIdx.x == 0) {
        unsigned long long t_end;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_end) :: "memory");
        if (blockIdx.x == 8) ((unsigned long long*)(out + 1536))[0] = (t_end - t_begin) / reps;   // s_memtime ticks per sweep
        out[blockIdx.x] = acc + m[1][1];
    }
#if ADKF_STAMP
    if (blockIdx.x == 3 && threadIdx.x < 128 && VAR == 0) ((unsigned long long*)(out + 1024))[threadIdx.x] = sm.stamp[threadIdx.x];
#endif
}

template <int VAR>
float run(const float* dA, float* dout, int T, int n, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_sweep<128, 512, VAR><<<T, 512>>>(dA, dout, n, 2);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k_sweep<128, 512, VAR><<<T, 512>>>(dA, dout, n, reps);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 256, n = 128, reps = 20;
    std::vector<float> A((size_t)T * n * n);
    for (int t = 0; t < T; ++t) for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
        float d = (float)(i - j); A[((size_t)t * n + i) * n + j] = 0.7f * expf(-d * d / 50.f) + (i == j ? 0.1f : 0.f);
    }
    float *dA, *dout; hipMalloc(&dA, A.size() * 4); hipMalloc(&dout, (T + 2048) * 4); hipMemset(dout, 0, (T + 2048) * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
#if ADKF_STAMP
    {
        std::vector<unsigned long long> st(128);
        k_sweep<128, 512, 0><<<T, 512>>>(dA, dout, n, 1); hipDeviceSynchronize();
        hipMemcpy(st.data(), dout + 1024, 128 * 8, hipMemcpyDeviceToHost);
        unsigned long long t0 = ~0ull; for (int w = 0; w < 8; ++w) if (st[w * 16] && st[w * 16] < t0) t0 = st[w * 16];
        for (int w = 0; w < 8; ++w) { printf("wave %d:", w); for (int s_ = 0; s_ < 11; ++s_) printf(" %6lld", st[w * 16 + s_] ? (long long)(st[w * 16 + s_] - t0) : -1ll); printf("\n"); }
    }
#endif
    const float full = run<0>(dA, dout, T, n, reps);
    unsigned long long ticks = 0;
    hipMemcpy(&ticks, dout + 1536, 8, hipMemcpyDeviceToHost);
    printf("T=%d  us per sweep:  full %.1f (%llu s_memtime ticks: %.2f ticks/ns) | update-only %.1f | barriers-only %.1f\n", T, full, ticks,
           ticks / (full * 1e3), run<1>(dA, dout, T, n, reps), run<3>(dA, dout, T, n, reps));
    return 0;
}
