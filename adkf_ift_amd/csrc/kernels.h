// The per-task (non-GEMM) kernels of the pipeline.  One workgroup per task unless noted.
#pragma once
#include "problems.h"
#include "inner.h"

namespace adkf {

// ---- column mean of the support features (gpytorch centres both operands by x1.mean) -----------------
// block = 64 columns (grid: ceil(d/64) x T).  16-byte path (d a multiple of 4, aligned rows): 16 lanes x float4 cover the 64
// columns, 16 row groups, every thread's loads are independent and issued eight deep (was: 4 row groups of 64 lanes, one dword
// per row, 32 rows per thread: 10.5 us at C2 for 33.5 MB); the dword path stays for everything else.
__global__ __launch_bounds__(256) void k_colmean(const float* Zs, const int32_t* n_s, int ns_ld, int d, float* mean, int T) {
    __shared__ float part[16][64];
    const int t = blockIdx.y;
    const int n = n_s ? n_s[t] : ns_ld;
    const float* Z = Zs + (size_t)t * ns_ld * d;
    const bool vec = (d & 3) == 0 && (reinterpret_cast<uintptr_t>(Zs) & 15) == 0;
    if (vec) {
        const int q = threadIdx.x & 15, g = threadIdx.x >> 4, c4 = blockIdx.x * 64 + q * 4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        if (c4 < d) {
            for (int i0 = g; i0 < n; i0 += 16 * 8) {
                float4 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {   // clamped row, discarded below: no branch around the loads
                    const int i = i0 + 16 * u;
                    v[u] = *reinterpret_cast<const float4*>(Z + (size_t)(i < n ? i : n - 1) * d + c4);
                }
#pragma unroll
                for (int u = 0; u < 8; ++u)
                    if (i0 + 16 * u < n) { s.x += v[u].x; s.y += v[u].y; s.z += v[u].z; s.w += v[u].w; }
            }
        }
        *reinterpret_cast<float4*>(&part[g][q * 4]) = s;
        __syncthreads();
        const int cl = threadIdx.x, c = blockIdx.x * 64 + cl;
        if (cl < 64 && c < d) {
            float r = 0.f;
#pragma unroll
            for (int k = 0; k < 16; ++k) r += part[k][cl];
            mean[(size_t)t * d + c] = n > 0 ? r / (float)n : 0.f;
        }
        return;
    }
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + cl;
    float s = 0.f;
    if (c < d)
        for (int i = g; i < n; i += 4) s += Z[(size_t)i * d + c];
    part[g][cl] = s;
    __syncthreads();
    if (g == 0 && c < d) {
        s = (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
        mean[(size_t)t * d + c] = n > 0 ? s / (float)n : 0.f;
    }
}

// ---- K1: squared distances in GEMM form on the fp32 MFMA: D2_ij = |x_i|^2 + |y_j|^2 - 2 x_i . y_j on the centred rows ----
struct ProbDist {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = true;
    static constexpr int NRED = 0;
    const float *X, *Y, *mean; const int32_t *n_x, *n_y; int x_ld, y_ld, d; bool symmetric; float* D2;
    int mx, my, i_base, j_base; const float *Xi, *Yi, *mu, *nxi, *nyi; float* Do; bool vec;
    __device__ bool setup(int t) {
        mx = n_x ? n_x[t] : x_ld; my = n_y ? n_y[t] : y_ld;
        Xi = X + (size_t)t * x_ld * d; Yi = Y + (size_t)t * y_ld * d; mu = mean + (size_t)t * d;
        nxi = nyi = nullptr; Do = D2 + (size_t)t * x_ld * y_ld;
        return mx > 0 && my > 0;
    }
    __device__ int M() const { return mx; } __device__ int N() const { return my; } __device__ int K() const { return d; }
    __device__ float a(int i, int k) const { return Xi[(size_t)i * d + k] - mu[k]; }
    __device__ float b(int k, int j) const { return Yi[(size_t)j * d + k] - mu[k]; }
    static constexpr int A_NRAW = 2, B_NRAW = 2;   // two-phase operand path (gemm.h): loads, then the centring
    __device__ bool raw_ok() const { return true; }
    __device__ void a_raw(int i, int k, float4 (&r)[2]) const { r[0] = ldq(Xi + (size_t)i * d + k); r[1] = ldq(mu + k); }
    __device__ void b_raw(int k, int j, float4 (&r)[2]) const { r[0] = ldq(Yi + (size_t)j * d + k); r[1] = ldq(mu + k); }
    __device__ void a_fin(int, int, const float4 (&r)[2], float (&v)[4]) const {
        v[0] = r[0].x - r[1].x; v[1] = r[0].y - r[1].y; v[2] = r[0].z - r[1].z; v[3] = r[0].w - r[1].w;
    }
    __device__ void b_fin(int k, int j, const float4 (&r)[2], float (&v)[4]) const { a_fin(j, k, r, v); }
    __device__ void a4(int i, int k, float (&v)[4]) const { float4 r[2]; a_raw(i, k, r); a_fin(i, k, r, v); }
    __device__ void b4(int k, int j, float (&v)[4]) const { float4 r[2]; b_raw(k, j, r); b_fin(k, j, r, v); }
    // symmetric (X == Y): only the tiles on or above the diagonal are computed; they also write their mirror image, so
    // the result is EXACTLY symmetric
    __device__ bool active(int m0, int n0) const { return !symmetric || m0 <= n0; }
    // squared norms of the centred rows: summed by the GEMM kernel itself while it stages the operands (gemm.h: set_rowsq),
    // so no separate pass over Z and no norm arrays
    __device__ void set_rowsq(const float* a, const float* b, int m0, int n0) { nxi = a; nyi = b; i_base = m0; j_base = n0; }
    __device__ float value(int i, int j, float acc) const {
        const float v = fmaxf(nxi[i - i_base] + nyi[j - j_base] - 2.f * acc, 0.f);
        return (symmetric && i == j) ? 0.f : v;
    }
    __device__ void epi(int i, int j, float acc, float*) const {
        const float v = value(i, j, acc);
        Do[(size_t)i * y_ld + j] = v;
        if (symmetric && (i / GT) < (j / GT)) Do[(size_t)j * y_ld + i] = v;
    }
    __device__ void epi4(int i0, int j, const float (&acc)[4], float* red) const {   // mirror image as ONE 16-byte store
        if (!(symmetric && vec && (i0 / GT) < (j / GT))) {
#pragma unroll
            for (int r = 0; r < 4; ++r) epi(i0 + r, j, acc[r], red);
            return;
        }
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = value(i0 + r, j, acc[r]); Do[(size_t)(i0 + r) * y_ld + j] = v[r]; }
        *reinterpret_cast<float4*>(Do + (size_t)j * y_ld + i0) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __device__ void store_red(int, const float*) const {}
};

// Up to three distance blocks (support-support, query-support, query-query) in ONE launch: each block alone is 3-4 tiles
// per task, which fills the chip for less than one round of resident workgroups and leaves launch, setup, first-fetch
// and epilogue latency uncovered (three launches: 3 x 33 us at C2; one launch of all ten tiles per task: see DESIGN.md).
// gemm.h's optional select() hook maps the flat tile index to (block, tile within the block).
struct ProbDistMulti {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = true;
    static constexpr int NRED = 0;
    ProbDist s0, s1, s2; int end0, end1, tn0, tn1, tn2;   // tiles [0, end0) -> s0, [end0, end1) -> s1, the rest -> s2
    ProbDist q; bool vec;
    __device__ void select(int& tile, int& tiles_n) {
        if (tile < end0) { q = s0; tiles_n = tn0; }
        else if (tile < end1) { q = s1; tiles_n = tn1; tile -= end0; }
        else { q = s2; tiles_n = tn2; tile -= end1; }
    }
    __device__ bool setup(int t) { return q.setup(t); }
    __device__ int M() const { return q.M(); } __device__ int N() const { return q.N(); } __device__ int K() const { return q.K(); }
    __device__ float a(int i, int k) const { return q.a(i, k); }
    __device__ float b(int k, int j) const { return q.b(k, j); }
    __device__ void a4(int i, int k, float (&v)[4]) const { q.a4(i, k, v); }
    __device__ void b4(int k, int j, float (&v)[4]) const { q.b4(k, j, v); }
    static constexpr int A_NRAW = 2, B_NRAW = 2;
    __device__ bool raw_ok() const { return true; }
    __device__ void a_raw(int i, int k, float4 (&r)[2]) const { q.a_raw(i, k, r); }
    __device__ void b_raw(int k, int j, float4 (&r)[2]) const { q.b_raw(k, j, r); }
    __device__ void a_fin(int i, int k, const float4 (&r)[2], float (&v)[4]) const { q.a_fin(i, k, r, v); }
    __device__ void b_fin(int k, int j, const float4 (&r)[2], float (&v)[4]) const { q.b_fin(k, j, r, v); }
    __device__ bool active(int m0, int n0) const { return q.active(m0, n0); }
    __device__ void epi(int i, int j, float acc, float* red) const { q.epi(i, j, acc, red); }
    __device__ void epi4(int i0, int j, const float (&acc)[4], float* red) const { q.epi4(i0, j, acc, red); }
    __device__ void set_rowsq(const float* a, const float* b, int m0, int n0) { q.set_rowsq(a, b, m0, n0); }
    __device__ void store_red(int, const float*) const {}
};

// ---- a4: fresh phi and priors ---------------------------------------------------------------------------
struct InitArgs { int numeric, use_ls_prior; float* phi; float* priors; };   // phi == null: nothing to initialise

__device__ __forceinline__ void init_params_task(const InitArgs& a, int t, float l0) {
    const float scale = 0.25f;
    const float mode = a.numeric ? 0.01f : 0.1f;
    a.phi[t * 3 + 0] = inv_softplus_f(mode - NOISE_LB);
    a.phi[t * 3 + 1] = 0.f;
    a.phi[t * 3 + 2] = inv_softplus_f(l0);
    a.priors[t * 4 + 0] = logf(mode) + scale * scale;
    a.priors[t * 4 + 1] = scale;
    a.priors[t * 4 + 2] = a.use_ls_prior ? logf(l0) + scale * scale : 0.f;
    a.priors[t * 4 + 3] = a.use_ls_prior ? scale : -1.f;
}

__global__ void k_init_params(const float* l0, int T, InitArgs a) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T) init_params_task(a, t, l0[t]);
}

// ---- K10: median heuristic.  Exact lower median of the positive strict-upper-triangle entries by a
// 31-step radix select on the float bit patterns (positive floats order like their bits). ---------------
// One workgroup per task; the candidates are loaded ONCE into registers, every radix step is then one compare per held
// value whose wave-wide count is the population count of the compare mask (a scalar instruction, no lane shuffles) and ONE
// barrier for the cross-wave sum (partials alternate between two LDS rows).  Only the strict upper triangle is held:
// rows i and n-1-i are folded into one row of n-1 candidates, so ceil(n/2) (n-1) slots cover all n (n-1) / 2 pairs.
template <int NT, int EPT>  // NT * EPT >= ceil(ld / 2) * (ld - 1): <512, 16> up to 128 points, <1024, 32> up to 256
__global__ __launch_bounds__(NT) void k_median(const float* D2ss, const int32_t* n_s, int ld, float* l0, int T, InitArgs init) {
    constexpr int NW = NT / 64;
    __shared__ int red[2][NW];
    int t, tile;
    if (!task_tile(T, 1, t, tile)) return;
    const int n = n_s ? n_s[t] : ld;
    const uint32_t* D = reinterpret_cast<const uint32_t*>(D2ss + (size_t)t * ld * ld);
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (n <= 1 || n > ld) {   // an empty, single-point (or corrupt) task: no candidates
        if (tid == 0) { l0[t] = 0.f; if (init.phi) init_params_task(init, t, 0.f); }
        return;
    }
    const int nm1 = n - 1, slots = ((n + 1) >> 1) * nm1;
    uint32_t v[EPT];
    int nonzero = 0;   // wave-uniform
#pragma unroll
    for (int r = 0; r < EPT; ++r) {
        const int e = r * NT + tid;
        // entries are clamped >= 0: 0 marks "not a candidate".  The load itself is unconditional (slot clamped into range, value
        // discarded): a branch around it would give each of the EPT loads its own basic block and its own s_waitcnt
        const int ec = e < slots ? e : slots - 1;
        const int fr = ec / nm1, c = ec - fr * nm1, top = nm1 - fr;     // row fr holds `top` candidates, row n-1-fr holds fr
        const int i = c < top ? fr : top, j = c < top ? fr + 1 + c : top + 1 + (c - top);
        uint32_t val = D[(size_t)i * ld + j];
        if (!(e < slots && (c < top || top != fr))) val = 0u;          // odd n: the middle row is folded onto itself
        v[r] = val;
        nonzero += __popcll(__ballot(val != 0u));
    }
    const int zeros = EPT * 64 - nonzero;
    if (lane == 0) red[0][wv] = nonzero;
    __syncthreads();
    int total = 0;
#pragma unroll
    for (int i = 0; i < NW; ++i) total += red[0][i];
    if (total == 0) { if (tid == 0) { l0[t] = 0.f; if (init.phi) init_params_task(init, t, 0.f); } return; }
    int rank = (total - 1) / 2;  // torch.median: lower median
    uint32_t prefix = 0;
    int buf = 1;
    for (int bit = 30; bit >= 0; --bit, buf ^= 1) {
        const uint32_t hi_mask = ~((1u << bit) - 1u);  // bits >= bit
        int c0 = 0;
#pragma unroll
        for (int r = 0; r < EPT; ++r) c0 += __popcll(__ballot((v[r] & hi_mask) == prefix));  // prefix matches, this bit = 0
        if (prefix == 0u) c0 -= zeros;   // the non-candidates match the all-zero prefix
        if (lane == 0) red[buf][wv] = c0;
        __syncthreads();   // one barrier per step: row `buf` was last read two steps ago, before the previous barrier
        c0 = 0;
#pragma unroll
        for (int i = 0; i < NW; ++i) c0 += red[buf][i];
        if (rank >= c0) { rank -= c0; prefix |= (1u << bit); }
    }
    if (tid == 0) {
        const float l = sqrtf(0.5f * __uint_as_float(prefix));
        l0[t] = l;
        if (init.phi) init_params_task(init, t, l);   // a4 in the same launch
    }
}

#if ADKF_STAMP_SMALL   // tools/small_bench.hip: s_memtime of the phases of the per-task kernels (workgroup 3, thread 0)
__device__ unsigned long long g_small_stamps[32];
#define ADKF_SST(k_) do { if (blockIdx.x == 3 && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); g_small_stamps[k_] = t_; } } while (0)
#else
#define ADKF_SST(k_) do {} while (0)
#endif

// ---- Stage C: beta = G alpha, gamma = Ainv alpha, delta = Ainv beta, traces, 3x3 Hessian ------------------
// (oracle/closed_form.py::inner_stage, want_hessian branch)
struct HessArgs { TaskView tv; const float* Ainv; const float* P; const float* D2ss; const float* y_s; const float* priors; float* scal; float* vecs; int T; };

constexpr int SMALL_NT = 1024;  // the per-task elementwise/mat-vec kernels: 16 waves per task

// The analytic 3x3 Hessian of f_inner in the raw parameters from the nine reductions
// acc = {tr(Ainv^2), tr(P Ainv), tr(P P), tr(Ainv K_ll), a^T K_ll a, a^T g, b^T g, b^T d, a^T b}
// (a = alpha, b = beta = G alpha, g = gamma = Ainv alpha, d = delta = Ainv beta) and the scalars of the evaluation.
__device__ __forceinline__ void hess_assemble(float* sc, const float* pri, int n, const float* acc) {
    const float noise = sc[S_NOISE], os = sc[S_OS], ls = sc[S_LS];
    const float trA2 = acc[0], trPA = acc[1], trPP = acc[2], trAinvKll = acc[3], aKlla = acc[4], ag = acc[5], bg = acc[6], bd = acc[7], ab = acc[8];
    const float trAinv = sc[S_TRAINV], aa = sc[S_AA], ya = sc[S_YA], trAinvG = sc[S_TRAINVG], aGa = sc[S_AGA];
    const float fn = (float)n;
    float h00 = ag - 0.5f * trA2;
    const float h01 = ((aa - noise * ag) - 0.5f * (trAinv - noise * trA2)) / os;
    const float h02 = bg - 0.5f * trPA;
    const float h11 = ((ya - 2.f * noise * aa + noise * noise * ag) - 0.5f * (fn - 2.f * noise * trAinv + noise * noise * trA2)) / (os * os);
    const float h12 = ((ab - noise * bg) - 0.5f * (trAinvG - noise * trPA)) / os - (0.5f * aGa - 0.5f * trAinvG) / os;
    float h22 = bd - 0.5f * aKlla - 0.5f * trPP + 0.5f * trAinvKll;
    // prior curvature (oracle/closed_form.py::lognormal_terms d2)
    if (pri[1] > 0.f) { const float lx = logf(noise), s2 = pri[1] * pri[1]; h00 -= (1.f + (lx - pri[0]) / s2 - 1.f / s2) / (noise * noise); }
    if (pri[3] > 0.f) { const float lx = logf(ls), s2 = pri[3] * pri[3]; h22 -= (1.f + (lx - pri[2]) / s2 - 1.f / s2) / (ls * ls); }
    const float d1[3] = {sc[S_D1N], sc[S_D1S], sc[S_D1L]}, d2[3] = {sc[S_D2N], sc[S_D2S], sc[S_D2L]};
    const float gt[3] = {sc[S_GT0], sc[S_GT1], sc[S_GT2]};
    const float h[3][3] = {{h00, h01, h02}, {h01, h11, h12}, {h02, h12, h22}};
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) sc[S_H0 + i * 3 + j] = (h[i][j] * d1[i] * d1[j] + (i == j ? gt[i] * d2[i] : 0.f)) / fn;
}

__global__ __launch_bounds__(SMALL_NT) void k_hess(HessArgs a) {
    constexpr int NT = SMALL_NT, NW = NT / 64;
    __shared__ float red[9 * NW];
    int t, tile;
    if (!task_tile(a.T, 1, t, tile)) return;
    const int n = a.tv.ns(t), ld = a.tv.ns_ld, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    float* sc = a.scal + (size_t)t * NSCAL;
    const float os = sc[S_OS], ls = sc[S_LS], il2 = 1.f / (ls * ls);
    const float* Ai = a.Ainv + (size_t)t * ld * ld;
    const float* Pi = a.P + (size_t)t * ld * ld;
    const float* D2 = a.D2ss + (size_t)t * ld * ld;
    float* al = a.vecs + ((size_t)t * NVEC + V_ALPHA) * a.tv.vld;
    float* be = a.vecs + ((size_t)t * NVEC + V_BETA) * a.tv.vld;
    float* ga = a.vecs + ((size_t)t * NVEC + V_GAMMA) * a.tv.vld;
    float* de = a.vecs + ((size_t)t * NVEC + V_DELTA) * a.tv.vld;
    const int kind = a.tv.kind;
    ADKF_SST(0);
    // wave per row: beta_i = sum_j G_ij alpha_j ; gamma_i = sum_j Ainv_ij alpha_j
    // (RF rows in flight per wave in both mat-vec passes: a row at a time is a chain of L2 round trips; RF = 8 makes each
    // pass ONE round of loads at 128 points)
    constexpr int RF = 8;
    for (int i0 = wv; i0 < n; i0 += RF * NW) {
        float sb[RF], sg[RF];
#pragma unroll
        for (int q = 0; q < RF; ++q) { sb[q] = 0.f; sg[q] = 0.f; }
        for (int j = lane; j < n; j += 64) {
            const float aj = al[j];
            float d2v[RF], av[RF];
#pragma unroll
            for (int q = 0; q < RF; ++q) {   // rows beyond n: clamped address, result discarded (no branch around a load)
                const int i = i0 + q * NW, ic = i < n ? i : n - 1;
                d2v[q] = D2[(size_t)ic * ld + j];
                av[q] = Ai[(size_t)ic * ld + j];
            }
#pragma unroll
            for (int q = 0; q < RF; ++q) {
                float k0, k1, k2; const float u = d2v[q] * il2; kappa3(kind, u, k0, k1, k2);
                const bool in = i0 + q * NW < n;
                sb[q] += in ? os * k1 * u * (-2.f / ls) * aj : 0.f;
                sg[q] += in ? av[q] * aj : 0.f;
            }
        }
#pragma unroll
        for (int q = 0; q < RF; ++q) {
            const int i = i0 + q * NW;
            const float b_ = wave_sum(sb[q]), g_ = wave_sum(sg[q]);
            if (lane == 0 && i < n) { be[i] = b_; ga[i] = g_; }
        }
    }
    __threadfence_block();
    __syncthreads();
    ADKF_SST(1);
    for (int i0 = wv; i0 < n; i0 += RF * NW) {
        float sd[RF];
#pragma unroll
        for (int q = 0; q < RF; ++q) sd[q] = 0.f;
        for (int j = lane; j < n; j += 64) {
            const float bj = be[j];
            float av[RF];
#pragma unroll
            for (int q = 0; q < RF; ++q) { const int i = i0 + q * NW; av[q] = Ai[(size_t)(i < n ? i : n - 1) * ld + j]; }
#pragma unroll
            for (int q = 0; q < RF; ++q) sd[q] += (i0 + q * NW < n) ? av[q] * bj : 0.f;
        }
#pragma unroll
        for (int q = 0; q < RF; ++q) {
            const int i = i0 + q * NW;
            const float d_ = wave_sum(sd[q]);
            if (lane == 0 && i < n) de[i] = d_;
        }
    }
    __threadfence_block();
    __syncthreads();
    ADKF_SST(2);
    // elementwise traces, by 32 x 32 tiles: thread (ty, tx) of tile (ib, jb) owns element (i, j) = (32 ib + ty, 32 jb + tx); the
    // partner P_ji of tr(P P) comes from tile (jb, ib), loaded row-wise as well and turned through LDS (read straight from
    // memory it is a column walk: 64 cache lines per wave instruction, the whole phase was 26 k cycles, 12 us).  At most
    // 16 tiles (this kernel serves up to 128 points): every load of every tile is issued before the first is used, to
    // clamped addresses where the tile sticks out of the matrix (no branch around a load), then one barrier per tile.
    float acc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};  // trA2, trPA, trPP, trAinvKll, aKlla, ag, bg, bd, ab
    {
        constexpr int TS = 32, SIDE = 4, MAXT = SIDE * SIDE;
        static_assert(NT == TS * TS, "one thread per tile element");
        __shared__ float turn[2][TS][TS + 1];
        const int ty = tid >> 5, tx = tid & 31, nside = (n + TS - 1) / TS;
        constexpr int HALF = MAXT / 4;   // four rounds of four tiles: 20 values in flight per thread (a 1024-thread workgroup has 128 registers per lane)
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            float va[HALF], vp[HALF], vt[HALF], vd[HALF], vaa[HALF];
#pragma unroll
            for (int kk = 0; kk < HALF; ++kk) {
                const int k = h * HALF + kk, ib = k / SIDE, jb = k % SIDE;
                const int i = min(ib * TS + ty, n - 1), j = min(jb * TS + tx, n - 1);      // my element
                const int it = min(jb * TS + ty, n - 1), jt = min(ib * TS + tx, n - 1);    // my element of the partner tile
                va[kk] = Ai[(size_t)i * ld + j]; vp[kk] = Pi[(size_t)i * ld + j]; vd[kk] = D2[(size_t)i * ld + j];
                vt[kk] = Pi[(size_t)it * ld + jt];
                vaa[kk] = al[i] * al[j];
            }
#pragma unroll
            for (int kk = 0; kk < HALF; ++kk) {
                const int k = h * HALF + kk, ib = k / SIDE, jb = k % SIDE;
                if (ib < nside && jb < nside) {     // workgroup-uniform
                    turn[k & 1][ty][tx] = vt[kk];   // = P[32 jb + ty][32 ib + tx]
                    __syncthreads();                // (one barrier per tile: buffer k & 1 was last read before the previous barrier)
                    const float pji = turn[k & 1][tx][ty];   // = P[32 jb + tx][32 ib + ty] = P_ji; bank (tx + ty) mod 32: conflict-free
                    if (ib * TS + ty < n && jb * TS + tx < n) {
                        float k0, k1, k2; const float u = vd[kk] * il2; kappa3(kind, u, k0, k1, k2);
                        const float Kll = os * (k2 * 4.f * u * u + k1 * 6.f * u) * il2;
                        acc[0] += va[kk] * va[kk]; acc[1] += vp[kk] * va[kk]; acc[2] += vp[kk] * pji; acc[3] += va[kk] * Kll; acc[4] += vaa[kk] * Kll;
                    }
                }
            }
        }
    }
    if (tid < n) { acc[5] = al[tid] * ga[tid]; acc[6] = be[tid] * ga[tid]; acc[7] = be[tid] * de[tid]; acc[8] = al[tid] * be[tid]; }
    for (int i = tid + NT; i < n; i += NT) { acc[5] += al[i] * ga[i]; acc[6] += be[i] * ga[i]; acc[7] += be[i] * de[i]; acc[8] += al[i] * be[i]; }
    ADKF_SST(3);
    block_sum<9, NT>(acc, red);
    ADKF_SST(4);
    if (tid == 0) hess_assemble(sc, a.priors + t * 4, n, acc);
}

// ---- alpha += A^-1 (y - A alpha): the same refinement step as ProbCres / ProbCfix for the one right-hand side y ------------
struct AlphaRefineArgs { TaskView tv; const float* Ainv; const float* D2ss; const float* y_s; float* vecs; float thresh; int T; };

__global__ __launch_bounds__(SMALL_NT) void k_alpha_refine(AlphaRefineArgs a) {
    constexpr int NT = SMALL_NT, NW = NT / 64;
    __shared__ float res[256];
    int t, tile;
    if (!task_tile(a.T, 1, t, tile)) return;
    const int n = a.tv.ns(t), ld = a.tv.ns_ld, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const float* sc = a.tv.scal + (size_t)t * NSCAL;
    if (ld > 256 || (ld <= 128 && !(sc[S_CONDA] > a.thresh))) return;     // (uniform; beyond 256 points the blocked path keeps its alpha)
    if (sc[S_AREF] != 0.f) return;                                         // this alpha has had its step (an earlier call on the same batch with REUSE_INNER)
    const float os = sc[S_OS], noise = sc[S_NOISE], il2 = 1.f / (sc[S_LS] * sc[S_LS]);
    const float* Ai = a.Ainv + (size_t)t * ld * ld;
    const float* D2 = a.D2ss + (size_t)t * ld * ld;
    const float* y = a.y_s + (size_t)t * ld;
    float* al = a.vecs + ((size_t)t * NVEC + V_ALPHA) * a.tv.vld;
    const int kind = a.tv.kind;
    for (int i = wv; i < n; i += NW) {     // residual, wave per row
        float s = 0.f;
        for (int j = lane; j < n; j += 64) s += (os * kappa0(kind, D2[(size_t)i * ld + j] * il2) + (i == j ? noise : 0.f)) * al[j];
        s = wave_sum(s);
        if (lane == 0) res[i] = y[i] - s;
    }
    __syncthreads();
    for (int i = wv; i < n; i += NW) {
        float s = 0.f;
        for (int j = lane; j < n; j += 64) s += Ai[(size_t)i * ld + j] * res[j];
        s = wave_sum(s);
        if (lane == 0) al[i] += s;
    }
    if (tid == 0) const_cast<float*>(sc)[S_AREF] = 1.f;
}

// ---- Stage D core: mu = C y, r = y_q - mu, factor S, e = S^-1 r, f_out, Cte = C^T e --------------------
struct OuterArgs { TaskView tv; const float* C; float* S; const float* y_s; const float* y_q; float* vecs; float* scal; float* f_out; int32_t* info; int T;
                   int reset_info; };  // reset_info: info[] holds nothing yet (the inner stage was reused): write, do not merge

template <int NMAX, int NT>
__global__ __launch_bounds__(NT) void k_outer_factor(OuterArgs a) {
    using SW = Sweep<NMAX, NT>;
    constexpr int RB = SW::RB, CB = SW::CB;
    __shared__ SweepSmem<NMAX, NT> sm;
    int t, tile;
    if (!task_tile(a.T, 1, t, tile)) return;
    const int n = a.tv.ns(t), m = a.tv.nq(t), tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    constexpr int NW = NT / 64;
    const float* Ci = a.C + (size_t)t * a.tv.nq_ld * a.tv.ns_ld;
    float* Si = a.S + (size_t)t * a.tv.nq_ld * a.tv.nq_ld;
    const float* ys = a.y_s + (size_t)t * a.tv.ns_ld;
    const float* yq = a.y_q + (size_t)t * a.tv.nq_ld;
    float* vbase = a.vecs + (size_t)t * NVEC * a.tv.vld;
    const int j0 = SW::bc() * CB;
    if (tid < NMAX) sm.vec_in[tid] = 0.f;
    __syncthreads();
    ADKF_SST(0);
    // this thread's block of S (exactly symmetric by construction), identity-padded: the loads are issued first and
    // land while the residual below is formed
    float mm[RB][CB];
    const bool s_vec = rows_aligned16(Si, a.tv.nq_ld);
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int i = SW::row(r);
        load_segment<CB>(Si + (size_t)i * a.tv.nq_ld, j0, m, i < m, s_vec, mm[r]);   // ProbS mirrors its tiles
#pragma unroll
        for (int c = 0; c < CB; ++c)
            if (i == j0 + c && i >= m) mm[r][c] = 1.f;                                 // identity padding
    }
    // residual r = y_q - C y_s  (wave per row; ALL of a wave's rows in flight at once: a row at a time is a chain of L2
    // round trips - 17 us at 128 x 128, measured; four at a time left four trips)
    {
        constexpr int RW = NMAX / NW;   // rows per wave
        float s[RW];
#pragma unroll
        for (int u = 0; u < RW; ++u) s[u] = 0.f;
        for (int j = lane; j < (m > 0 ? n : 0); j += 64) {
            const float yj = ys[j];
            // (rows beyond m: the load goes to row m - 1 and is discarded - a wave-uniform branch per row would put every
            // load in its own basic block with its own s_waitcnt: 32 serial round trips instead of 2, measured 22 k cycles)
            float cv[RW];
#pragma unroll
            for (int u = 0; u < RW; ++u) { const int i = wv + u * NW; cv[u] = Ci[(size_t)(i < m ? i : m - 1) * a.tv.ns_ld + j]; }
#pragma unroll
            for (int u = 0; u < RW; ++u) { const int i = wv + u * NW; s[u] += i < m ? cv[u] * yj : 0.f; }
        }
#pragma unroll
        for (int u = 0; u < RW; ++u) {
            const int i = wv + u * NW;
            const float t_ = wave_sum(s[u]);
            if (lane == 0 && i < m) { vbase[V_MU * a.tv.vld + i] = t_; sm.vec_in[i] = yq[i] - t_; }
        }
    }
    __syncthreads();
    ADKF_SST(1);
    SW::run(mm, m, sm);
    ADKF_SST(2);
    float logdet;
    const int info = SW::finish(m, sm, logdet);
    const float pivr = pivot_ratio<NT>(sm.pivs, m, sm.red);
    ADKF_SST(3);
    if (tid == 0) a.scal[(size_t)t * NSCAL + S_PIVR_S] = pivr;
    SW::solve(mm, sm.vec_in, sm.vec_out);  // e = S^-1 r
    ADKF_SST(4);
    float q[1] = {0.f};
    if (tid < m) {
        const float e = sm.vec_out[tid], r = sm.vec_in[tid];
        vbase[V_E * a.tv.vld + tid] = e;
        vbase[V_R * a.tv.vld + tid] = r;
        q[0] = r * e;
    }
    block_sum<1, NT>(q, sm.red);
    ADKF_SST(5);
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int i = SW::row(r);
        float neg[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c) neg[c] = -mm[r][c];
        store_segment<CB>(Si + (size_t)i * a.tv.nq_ld, j0, m, i < m, s_vec, neg);
    }
    ADKF_SST(6);
    // Cte_j = sum_i C_ij e_i  (thread per column: coalesced; the rows are split over PARTS thread groups so that every
    // thread works and a column is PARTS short chains of loads instead of one long one: 12 -> 4 dependent round trips
    // at 128 x 128)
    {
        constexpr int PARTS = NMAX >= 64 ? NT / NMAX : 1, COLS = NT / PARTS;
        static_assert(PARTS == 1 || SweepSmem<NMAX, NT>::SCRATCH_FLOATS >= NT, "the partial sums reuse the sweep's LDS slots");
        float* part_s = sm.scratch();
        const int jl = tid % COLS, part = tid / COLS;
        const int per = (m + PARTS - 1) / PARTS, i_lo = part * per, i_hi = min(m, i_lo + per);
        for (int jb = 0; jb < n; jb += COLS) {   // n is the SUPPORT count: it may exceed NMAX (which follows the query count)
            const int j = jb + jl;
            float s8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            if (j < n) {
                int i = i_lo;
                for (; i + 32 <= i_hi; i += 32) {   // a part of 32 rows (128 points, four parts) is ONE round of loads
                    float cv[32];
#pragma unroll
                    for (int u = 0; u < 32; ++u) cv[u] = Ci[(size_t)(i + u) * a.tv.ns_ld + j];
#pragma unroll
                    for (int u = 0; u < 32; ++u) s8[u & 7] += cv[u] * sm.vec_out[i + u];
                }
                for (; i + 8 <= i_hi; i += 8) {
#pragma unroll
                    for (int u = 0; u < 8; ++u) s8[u] += Ci[(size_t)(i + u) * a.tv.ns_ld + j] * sm.vec_out[i + u];
                }
                for (; i < i_hi; ++i) s8[0] += Ci[(size_t)i * a.tv.ns_ld + j] * sm.vec_out[i];
            }
            const float part_sum = ((s8[0] + s8[1]) + (s8[2] + s8[3])) + ((s8[4] + s8[5]) + (s8[6] + s8[7]));
            if (PARTS == 1) {
                if (j < n) vbase[V_CTE * a.tv.vld + j] = part_sum;
            } else {
                __syncthreads();
                part_s[tid] = part_sum;
                __syncthreads();
                if (part == 0 && j < n) {
                    float c = 0.f;
#pragma unroll
                    for (int q2 = 0; q2 < PARTS; ++q2) c += part_s[q2 * COLS + jl];
                    vbase[V_CTE * a.tv.vld + j] = c;
                }
            }
        }
    }
    ADKF_SST(7);
    if (tid == 0) {
        const float f = 0.5f * q[0] + 0.5f * logdet + 0.5f * (float)m * LOG_2PI;
        a.scal[(size_t)t * NSCAL + S_FOUT] = f;
        a.scal[(size_t)t * NSCAL + S_LOGDETS] = logdet;
        if (a.f_out) a.f_out[t] = (info == 0) ? f : NAN;
        if (a.reset_info) a.info[t] = info != 0 ? 100000 + info : 0;
        else if (info != 0 && a.info[t] == 0) a.info[t] = 100000 + info;
    }
}

// ---- W_ss for d f_in / dZ:  Q . s kappa'/l^2,  Q = (Ainv - alpha alpha^T) / (2n) -------------------------
struct WinArgs { TaskView tv; const float* Ainv; const float* D2ss; float* Wss; const float* scal; int T; int tiles; };

// `tiles` workgroups per task (one per task was 46 us at 256 x 128^2 - an element-wise pass should not be latency-bound)
__global__ __launch_bounds__(256) void k_win(WinArgs a) {
    int t, tile;
    if (!task_tile(a.T, a.tiles, t, tile)) return;
    const int n = a.tv.ns(t), ld = a.tv.ns_ld;
    const float* sc = a.scal + (size_t)t * NSCAL;
    const float os = sc[S_OS], ls = sc[S_LS], il2 = 1.f / (ls * ls);
    const float* Ai = a.Ainv + (size_t)t * ld * ld;
    const float* D2 = a.D2ss + (size_t)t * ld * ld;
    const float* al = a.tv.vec_ptr(t, V_ALPHA);
    float* Wo = a.Wss + (size_t)t * ld * ld;
    const float cf = 0.5f / (float)n * os * il2;
    for (int e = tile * 256 + threadIdx.x; e < n * n; e += 256 * a.tiles) {
        const int i = e / n, j = e - i * n;
        float k0, k1, k2; kappa3(a.tv.kind, D2[(size_t)i * ld + j] * il2, k0, k1, k2);
        Wo[(size_t)i * ld + j] = (Ai[(size_t)i * ld + j] - al[i] * al[j]) * cf * k1;
    }
}

// ---- g_out, v = H^-1 g_out, coefficients and the vector w = A^-1 B_v alpha -------------------------------
struct SolveArgs { TaskView tv; float* scal; float* vecs; const float* part_oc; const float* part_ma; int nt_oc, nt_ma; int flags; float* g_phi_out; float* v_out; float* H_out; int T; int with_hessian; };

// Called by ALL threads of a workgroup (any size >= 64); contains one barrier.
__device__ __forceinline__ void solve_v_task(const SolveArgs& a, int t) {
    float* sc = a.scal + (size_t)t * NSCAL;
    const int n = a.tv.ns(t), lane = threadIdx.x;
    __shared__ float sh[8];
    if (lane == 0) {
        double oc0 = 0.0, oc1 = 0.0, ma0 = 0.0, ma1 = 0.0, ma2 = 0.0;   // (the three pieces of each component nearly cancel: summed in float64)
        for (int q = 0; q < a.nt_oc; ++q) { oc0 += a.part_oc[((size_t)t * a.nt_oc + q) * 4 + 0]; oc1 += a.part_oc[((size_t)t * a.nt_oc + q) * 4 + 1]; }
        for (int q = 0; q < a.nt_ma; ++q) { const float* p = a.part_ma + ((size_t)t * a.nt_ma + q) * 4; ma0 += p[0]; ma1 += p[1]; ma2 += p[2]; }
        const float g_noise = (float)((double)sc[S_QQ_TR] + ma0);
        const float g_s = (float)(ma1 + oc0 + (double)sc[S_QQ_K]);
        const float g_l = (float)(ma2 + oc1 + (double)sc[S_QQ_L]);
        float g[3] = {g_noise * sc[S_D1N], g_s * sc[S_D1S], g_l * sc[S_D1L]};
        sc[S_GOUT0] = g[0]; sc[S_GOUT1] = g[1]; sc[S_GOUT2] = g[2];
        if (a.g_phi_out) { a.g_phi_out[t * 3 + 0] = g[0]; a.g_phi_out[t * 3 + 1] = g[1]; a.g_phi_out[t * 3 + 2] = g[2]; }
        float v[3] = {0.f, 0.f, 0.f};
        if (a.with_hessian && !(a.flags & 1)) {
            // 3x3 Gaussian elimination with partial pivoting (the reference: torch.linalg.solve,
            // fs_mol/utils/cauchy_hypergradient.py:136)
            double Mx[3][4];   // 3 x 3 elimination in float64 (free; cond(H) reaches 1e2 .. 1e3)
            for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) Mx[i][j] = sc[S_H0 + i * 3 + j]; Mx[i][3] = g[i]; }
            for (int c = 0; c < 3; ++c) {
                int pv = c;
                for (int r = c + 1; r < 3; ++r) if (fabs(Mx[r][c]) > fabs(Mx[pv][c])) pv = r;
                if (pv != c) for (int j = 0; j < 4; ++j) { const double tmp = Mx[c][j]; Mx[c][j] = Mx[pv][j]; Mx[pv][j] = tmp; }
                const double ip = 1.0 / Mx[c][c];
                for (int r = c + 1; r < 3; ++r) { const double f = Mx[r][c] * ip; for (int j = c; j < 4; ++j) Mx[r][j] -= f * Mx[c][j]; }
            }
            double vd[3] = {0.0, 0.0, 0.0};
            for (int c = 2; c >= 0; --c) { double s = Mx[c][3]; for (int j = c + 1; j < 3; ++j) s -= Mx[c][j] * vd[j]; vd[c] = s / Mx[c][c]; }
            for (int c = 0; c < 3; ++c) v[c] = (float)vd[c];
        }
        sc[S_V0] = v[0]; sc[S_V1] = v[1]; sc[S_V2] = v[2];
        const float cn = v[0] * sc[S_D1N], cs = v[1] * sc[S_D1S] / sc[S_OS], cl = v[2] * sc[S_D1L];
        sc[S_CN] = cn; sc[S_CS] = cs; sc[S_CL] = cl;
        sh[0] = cn; sh[1] = cs; sh[2] = cl; sh[3] = sc[S_NOISE];
        if (a.v_out) { a.v_out[t * 3 + 0] = v[0]; a.v_out[t * 3 + 1] = v[1]; a.v_out[t * 3 + 2] = v[2]; }
        if (a.H_out) for (int q = 0; q < 9; ++q) a.H_out[t * 9 + q] = a.with_hessian ? sc[S_H0 + q] : 0.f;
    }
    __syncthreads();
    if (a.with_hessian) {
        const float cn = sh[0], cs = sh[1], cl = sh[2], noise = sh[3];
        float* vb = a.vecs + (size_t)t * NVEC * a.tv.vld;
        for (int i = lane; i < n; i += (int)blockDim.x) {
            const float al = vb[V_ALPHA * a.tv.vld + i], ga = vb[V_GAMMA * a.tv.vld + i], de = vb[V_DELTA * a.tv.vld + i];
            vb[V_W * a.tv.vld + i] = cn * ga + cs * (al - noise * ga) + cl * de;
        }
    }
}

__global__ __launch_bounds__(64) void k_solve_v(SolveArgs a) {
    const int t = blockIdx.x;
    if (t >= a.T) return;
    solve_v_task(a, t);
}

// ---- W_qq = dir * Omega . s kappa'(u_qq)/l^2 and its three reductions -------------------------------------
struct WqqArgs { TaskView tv; const float* Sinv; const float* D2qq; float* Wqq; float* scal; float dirscale; int T;
                 int do_solve; SolveArgs solve; };  // do_solve: finish with solve_v_task (saves the k_solve_v launch)

__global__ __launch_bounds__(SMALL_NT) void k_wqq(WqqArgs a) {
    constexpr int NT = SMALL_NT;
    __shared__ float red[3 * (NT / 64)];
    int t, tile;
    if (!task_tile(a.T, 1, t, tile)) return;
    const int m = a.tv.nq(t), ld = a.tv.nq_ld, tid = threadIdx.x;
    float* sc = a.scal + (size_t)t * NSCAL;
    const float os = sc[S_OS], ls = sc[S_LS], il2 = 1.f / (ls * ls);
    const float* Si = a.Sinv + (size_t)t * ld * ld;
    const float* D2 = a.D2qq + (size_t)t * ld * ld;
    float* Wo = a.Wqq + (size_t)t * ld * ld;
    const float* ev = a.tv.vec_ptr(t, V_E);
    float acc[3] = {0.f, 0.f, 0.f};
    for (int e0 = tid; e0 < m * m; e0 += 4 * NT) {   // four elements per trip: their loads go out together
        float sv[4], dv[4], ee[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = min(e0 + q * NT, m * m - 1);   // clamped: no branch around the loads (the tail is skipped below)
            const int i = e / m, j = e - i * m;
            sv[q] = Si[(size_t)i * ld + j]; dv[q] = D2[(size_t)i * ld + j]; ee[q] = ev[i] * ev[j];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int e = e0 + q * NT;
            if (e >= m * m) break;
            const int i = e / m, j = e - i * m;
            const float om = 0.5f * (sv[q] - ee[q]);
            float k0, k1, k2; const float u = dv[q] * il2; kappa3(a.tv.kind, u, k0, k1, k2);
            Wo[(size_t)i * ld + j] = a.dirscale * om * os * k1 * il2;
            if (i == j) acc[0] += om;
            acc[1] += om * k0;
            acc[2] += om * os * k1 * u * (-2.f / ls);
        }
    }
    block_sum<3, NT>(acc, red);
    if (tid == 0) { sc[S_QQ_TR] = acc[0]; sc[S_QQ_K] = acc[1]; sc[S_QQ_L] = acc[2]; }
    if (a.do_solve) {
        __threadfence_block();
        __syncthreads();
        solve_v_task(a.solve, t);
    }
}

// ---- predictive variance diag: var_i = s - sum_j C_ij Kqs_ij + noise;  mean_i = sum_j C_ij y_j ------------
struct PredArgs { TaskView tv; const float* C; const float* D2qs; const float* y_s; float* mean; float* var; const float* scal; int T; };

__global__ __launch_bounds__(256) void k_predict(PredArgs a) {
    int t, tile;
    if (!task_tile(a.T, 1, t, tile)) return;
    const int n = a.tv.ns(t), m = a.tv.nq(t), lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const float* sc = a.scal + (size_t)t * NSCAL;
    const float os = sc[S_OS], il2 = 1.f / (sc[S_LS] * sc[S_LS]), noise = sc[S_NOISE];
    const float* Ci = a.C + (size_t)t * a.tv.nq_ld * a.tv.ns_ld;
    const float* D2 = a.D2qs + (size_t)t * a.tv.nq_ld * a.tv.ns_ld;
    const float* ys = a.y_s + (size_t)t * a.tv.ns_ld;
    for (int i = wv; i < a.tv.nq_ld; i += 4) {
        float s1 = 0.f, s2 = 0.f;
        if (i < m)
            for (int j = lane; j < n; j += 64) {
                const float c = Ci[(size_t)i * a.tv.ns_ld + j];
                s1 += c * ys[j];
                s2 += c * os * kappa0(a.tv.kind, D2[(size_t)i * a.tv.ns_ld + j] * il2);
            }
        s1 = wave_sum(s1); s2 = wave_sum(s2);
        if (lane == 0) {
            a.mean[(size_t)t * a.tv.nq_ld + i] = (i < m) ? s1 : 0.f;
            if (a.var) a.var[(size_t)t * a.tv.nq_ld + i] = (i < m) ? (os - s2 + noise) : 0.f;
        }
    }
}

}  // namespace adkf
