// gemm_x3.h against gemm.h on the distance problem (ProbDistMulti): error against a host float64 reference (maximum, and the MEAN
// SIGNED error of the inner products - a truncating accumulator would show as a bias) and time per launch.  Not part of the library.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 -I adkf_ift_amd/csrc tools/x3_bench.hip -o tools/x3_bench
//   tools/x3_bench [T n d spread]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "kernels.h"
#include "gemm_x3.h"
using namespace adkf;

int main(int argc, char** argv) {
    const int T = argc > 1 ? atoi(argv[1]) : 256, n = argc > 2 ? atoi(argv[2]) : 128, d = argc > 3 ? atoi(argv[3]) : 256;
    const float spread = argc > 4 ? atof(argv[4]) : 1.f;   // features = cluster centre (|.| ~ 1) + spread * noise
    std::vector<float> hs((size_t)T * n * d), hq((size_t)T * n * d);
    unsigned s = 12345u;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((s >> 8) & 0xffff) / 65536.f - 0.5f; };
    std::vector<float> centre(d);
    for (auto& c : centre) c = 2.f * rnd();
    for (size_t i = 0; i < hs.size(); ++i) { hs[i] = centre[i % d] + spread * rnd(); hq[i] = centre[i % d] + spread * rnd(); }
    constexpr int NV = 4;
    const char* names[NV] = {"f32 mfma 64/256  ", "x3 64 / 256      ", "x3 128 / 512     ", "x3 128 / 256     "};
    float *Zs, *Zq, *mean, *D[NV][3];
    hipMalloc(&Zs, hs.size() * 4); hipMalloc(&Zq, hq.size() * 4); hipMalloc(&mean, (size_t)T * d * 4);
    for (int v = 0; v < NV; ++v) for (int b = 0; b < 3; ++b) hipMalloc(&D[v][b], (size_t)T * n * n * 4);
    hipMemcpy(Zs, hs.data(), hs.size() * 4, hipMemcpyHostToDevice); hipMemcpy(Zq, hq.data(), hq.size() * 4, hipMemcpyHostToDevice);
    hipMemset(mean, 0, (size_t)T * d * 4);   // (no centring: the worst case for the GEMM form)
    ProbDistMulti pm[NV];
    int totals[NV];
    for (int v = 0; v < NV; ++v) {
        const int edge = v < 2 ? 64 : 128;
        const int tn = (n + edge - 1) / edge, tiles = tn * tn;
        totals[v] = 3 * tiles;
        ProbDist p;
        p.mean = mean; p.d = d; p.vec = (d % 4) == 0; p.n_x = nullptr; p.n_y = nullptr; p.x_ld = n; p.y_ld = n;
        pm[v].vec = p.vec;
        p.X = Zs; p.Y = Zs; p.symmetric = true; p.D2 = D[v][0]; pm[v].s0 = p;
        p.X = Zq; p.Y = Zs; p.symmetric = false; p.D2 = D[v][1]; pm[v].s1 = p;
        p.X = Zq; p.Y = Zq; p.symmetric = true; p.D2 = D[v][2]; pm[v].s2 = p;
        pm[v].tn0 = pm[v].tn1 = pm[v].tn2 = tn; pm[v].end0 = tiles; pm[v].end1 = 2 * tiles;
    }
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int v = 0; v < NV; ++v)
        for (int rep = 0; rep < 3; ++rep) {
            hipEventRecord(e0);
            const int total = totals[v];
            for (int i = 0; i < 20; ++i) {
                if (v == 0) k_bgemm<ProbDistMulti, GT><<<T * total, 256>>>(pm[0], T, 1, total);
                else if (v == 1) k_bgemm3<ProbDistMulti, 64, 256><<<T * total, 256>>>(pm[1], T, 1, total);
                else if (v == 2) k_bgemm3<ProbDistMulti, 128, 512><<<T * total, 512>>>(pm[2], T, 1, total);
                else k_bgemm3<ProbDistMulti, 128, 256><<<T * total, 256>>>(pm[3], T, 1, total);
            }
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            if (rep == 2) printf("%s  %.2f us per launch (%s)\n", names[v], ms * 1000 / 20, hipGetErrorString(hipGetLastError()));
        }
    // accuracy on the query-support block of the first tasks
    const int TC = T < 4 ? T : 4;
    std::vector<float> o[NV];
    for (int v = 0; v < NV; ++v) { o[v].resize((size_t)TC * n * n); hipMemcpy(o[v].data(), D[v][1], o[v].size() * 4, hipMemcpyDeviceToHost); }
    double mx[NV] = {}, bias[NV] = {}, scale = 0;
    size_t cnt = 0;
    for (int t = 0; t < TC; ++t)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) {
                double ref = 0, nx = 0, ny = 0;
                for (int k = 0; k < d; ++k) {
                    const double x = hq[((size_t)t * n + i) * d + k], y = hs[((size_t)t * n + j) * d + k];
                    ref += (x - y) * (x - y); nx += x * x; ny += y * y;
                }
                const double sc = nx + ny;
                for (int v = 0; v < NV; ++v) {
                    const double e = (o[v][((size_t)t * n + i) * n + j] - ref) / sc;
                    if (fabs(e) > mx[v]) mx[v] = fabs(e);
                    bias[v] += e;
                }
                scale += sc; ++cnt;
            }
    for (int v = 0; v < NV; ++v)
        printf("%s  D2_qs error / (|x|^2 + |y|^2): max %.3e  mean signed %.3e\n", names[v], mx[v], bias[v] / cnt);
    // the symmetric blocks: both forms must agree with each other to rounding
    return 0;
}
