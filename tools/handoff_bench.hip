// Floor of the hand-off of one block step: what one "publish -> s_waitcnt -> s_barrier -> wake up -> ds_read -> use" round costs on a
// CU with 8 waves, in s_memtime ticks.  Not part of the library.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 tools/handoff_bench.hip -o tools/handoff_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ __launch_bounds__(512) void k(float* out, unsigned long long* ticks, int iters) {
    __shared__ float buf[2][1024];
    const int tid = threadIdx.x, w = tid >> 6;
    float v = tid * 1e-3f;
    if (tid < 1024) { buf[0][tid] = v; buf[0][tid + 512] = v; buf[1][tid] = v; buf[1][tid + 512] = v; }
    __syncthreads();
    unsigned long long t0 = 0, t1 = 0;
    if (tid == 0) asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0) :: "memory");
    for (int it = 0; it < iters; ++it) {
        const int slot = it & 1;
        if (MODE >= 1) {   // one wave publishes (rotating), 16 bytes per lane
            if (w == (it & 7)) {
                if (MODE >= 5) {   // a dependent VALU chain of ~ the 4 x 4 inversion's length (4 rcp + 4 x 8 dependent ops)
#pragma unroll
                    for (int q = 0; q < 4; ++q) { float r = __builtin_amdgcn_rcpf(v + 2.f); r = fmaf(fmaf(-(v + 2.f), r, 1.f), r, r); v = fmaf(v, r, 0.5f); v = fmaf(v, r, 0.25f); v = fmaf(v, r, 0.125f); v = fmaf(v, r, 0.0625f); }
                }
                *reinterpret_cast<float4*>(&buf[slot ^ 1][(tid & 63) * 4]) = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
                if (MODE >= 4) {
#pragma unroll
                    for (int q = 1; q < 4; ++q) *reinterpret_cast<float4*>(&buf[slot ^ 1][256 * q + (tid & 63) * 4]) = make_float4(v, v + 1.f, v + 2.f, v + 3.f);
                }
            }
        }
        __syncthreads();
        if (MODE >= 2) {   // everybody reads what was published and depends on it
            float s = buf[slot ^ 1][(tid * 7) & 255];
            if (MODE >= 3) {
#pragma unroll
                for (int q = 1; q < 10; ++q) s += buf[slot ^ 1][(tid * 7 + 64 * q) & 1023];
            }
            v = v * 0.5f + s * 1e-3f;
        }
    }
    if (tid == 0) { asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1) :: "memory"); ticks[blockIdx.x] = (t1 - t0) / iters; }
    out[blockIdx.x * 512 + tid] = v;
}

int main() {
    const int T = 256, iters = 2000;
    float* out; unsigned long long* ticks;
    hipMalloc(&out, T * 512 * 4); hipMalloc(&ticks, T * 8);
    std::vector<unsigned long long> h(T);
    const char* names[6] = {"barrier only", "+ one wave stores 16 B/lane before it", "+ everybody reads 1 dword after it (dependent)", "+ 10 dword reads", "+ publisher stores 4 x 16 B/lane", "+ publisher runs a 4-rcp dependent chain first"};
    for (int mode = 0; mode < 6; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            switch (mode) {
                case 0: k<0><<<T, 512>>>(out, ticks, iters); break;
                case 1: k<1><<<T, 512>>>(out, ticks, iters); break;
                case 2: k<2><<<T, 512>>>(out, ticks, iters); break;
                case 3: k<3><<<T, 512>>>(out, ticks, iters); break;
                case 4: k<4><<<T, 512>>>(out, ticks, iters); break;
                case 5: k<5><<<T, 512>>>(out, ticks, iters); break;
            }
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), ticks, T * 8, hipMemcpyDeviceToHost);
        printf("mode %d: %5llu ticks per round   (%s)\n", mode, h[8], names[mode]);
    }
    return 0;
}
