// Ablation of the publish chain of factor.h (which piece of the critical path costs what).  Not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "factor.h"
using namespace adkf;

template <int LEVEL>
__global__ __launch_bounds__(512) void k_chain(const float* A, float* out, int reps) {
    using SW = SweepBlk<128, 512>;
    constexpr int RB = 8, CB = 4, B = 4, NBC = 32;
    __shared__ SweepSmemBlk<128, 512> sm;
    const int i0 = SW::br() * RB, j0 = SW::bc() * CB;
    float m[RB][CB];
    for (int r = 0; r < RB; ++r) for (int c = 0; c < CB; ++c) m[r][c] = A[(i0 + r) * 128 + j0 + c];
    float acc = 0.f;
    for (int it = 0; it < reps; ++it) {
        for (int q = 0; q < 32; ++q) {
            __syncthreads();
            const int kb = q / 2, buf = q & 1;
            const int plane = (kb * NBC + q) & 63;
            float D[B][B];
            if (LEVEL >= 1) {
                for (int a = 0; a < B; ++a) for (int b = 0; b <= a; ++b) {
                    D[a][b] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m[a][b]), plane)); D[b][a] = D[a][b]; }
            } else {
                for (int a = 0; a < B; ++a) for (int b = 0; b < B; ++b) D[a][b] = (a == b) ? 1.f : 0.1f;
            }
            if (SW::br() == kb) {
                float C[B][CB], F[B][CB], piv[B];
                for (int a = 0; a < B; ++a) for (int c = 0; c < CB; ++c) C[a][c] = m[a][c];
                if (LEVEL >= 2) InvSpd<B>::run(D, piv);
                if (LEVEL >= 3) {
                    for (int a = 0; a < B; ++a) for (int c = 0; c < CB; ++c) { float s = 0.f; for (int b = 0; b < B; ++b) s = fmaf(D[a][b], C[b][c], s); F[a][c] = s; }
                } else {
                    for (int a = 0; a < B; ++a) for (int c = 0; c < CB; ++c) F[a][c] = C[a][c] * D[a][a];
                }
                for (int a = 0; a < B; ++a) for (int c = 0; c < CB; ++c) { sm.cross[buf][a][j0 + c] = C[a][c]; sm.fvec[buf][a][j0 + c] = F[a][c]; }
                for (int a = 0; a < B; ++a) m[a][a] += 1e-6f * F[a][a];
            }
            if (LEVEL >= 4) {  // consumers read the vectors back after the next barrier
                __syncthreads();
                float s = 0.f;
                for (int a = 0; a < B; ++a) { for (int r = 0; r < RB; ++r) s += sm.fvec[buf][a][i0 + r]; for (int c = 0; c < CB; ++c) s += sm.cross[buf][a][j0 + c]; }
                m[7][3] += 1e-9f * s;
            }
        }
        acc += m[0][0];
    }
    if (threadIdx.x == 0) out[blockIdx.x] = acc + m[1][1] + m[7][3];
}

template <int L>
float run(const float* dA, float* dout, int T, int reps) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k_chain<L><<<T, 512>>>(dA, dout, 2); hipDeviceSynchronize();
    hipEventRecord(e0); k_chain<L><<<T, 512>>>(dA, dout, reps); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e6f / reps / 32;  // ns per block step
}
int main() {
    const int T = 256, n = 128;
    std::vector<float> A((size_t)n * n);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { float d = (float)(i - j); A[i * n + j] = 0.7f * expf(-d * d / 50.f) + (i == j ? 0.1f : 0.f); }
    float *dA, *dout; hipMalloc(&dA, A.size() * 4); hipMalloc(&dout, T * 4);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    printf("ns per block step: write-only %.0f | +readlane %.0f | +inverse %.0f | +F %.0f | +consumer read-back (2 barriers) %.0f\n",
           run<0>(dA, dout, T, 20), run<1>(dA, dout, T, 20), run<2>(dA, dout, T, 20), run<3>(dA, dout, T, 20), run<4>(dA, dout, T, 20));
    return 0;
}
