// Batched small-matrix GEMM on the fp32-input MFMA (v_mfma_f32_16x16x4_f32), one TM x TM output tile per
// 256-thread workgroup (TM = 64 or 128), operands produced ON THE FLY by a problem functor (kernel-matrix entries
// from squared distances, Omega from S^-1 and e, ...) and a fused epilogue functor (elementwise chain rule +
// reductions), so no intermediate kernel matrix is ever written to HBM.
//
// Pipeline: K is consumed in chunks of 32.  Tiles that lie fully inside their task take the two-phase operand path: the
// 16-byte global loads of chunk c+1 (x_raw) are issued before the MFMAs of chunk c, the functor arithmetic on them
// (x_fin: centring, exp, scaling) and the LDS stores come after, so no wave waits for memory in front of its matrix
// instructions.  Ragged tiles take the checked path (a4 / b4 or element-wise, range-checked per lane, loaded AND
// transformed before the MFMAs).  LDS layouts are conflict-free for the b32 fragment reads: [mn][34] for K-contiguous
// operands (bank = 2 i + k), [k][TM + 16] for MN-contiguous ones.
//
// Tile size: the 128 x 128 tile halves the L2 -> CU operand traffic of the 64 x 64 one, but with 256 threads it leaves
// one wave per SIMD and nothing to cover the stage/barrier phases: measured 1.4-1.8x SLOWER on every C2 stage, so the
// host (adkf_gp.hip::tile_edge) always picks 64; the variant stays for experiments with more waves per tile.
// Also measured without gain: a double-buffered LDS tile with one barrier per chunk (1.530 -> 1.549 ms per C2 step:
// the second buffer halves the workgroups per CU, which costs more than the removed barrier).
//
// A problem type P provides
//   static constexpr bool A_KCONTIG / B_KCONTIG : is the operand contiguous in memory along k?  (chooses
//                                                  the coalesced thread->element map and the LDS layout)
//   static constexpr int  NRED                   : per-tile reductions the epilogue accumulates
//   __device__ bool  setup(int task)             : loads per-task sizes/scalars; false = nothing to do
//   __device__ int   M(), N(), K()               : per-task logical sizes
//   __device__ float a(int i, int k), b(int k, int j)   : operand entries (callers guarantee in-range)
//   __device__ void  epi(int i, int j, float acc, float* red)
//   __device__ void  store_red(int tile, const float* red)   (only when NRED > 0; called by thread 0)
//   __device__ bool  skip(int m0, int n0)        : OPTIONAL - true when the tile at (m0, n0) needs no product (its
//                                                  epilogue still runs, with acc = 0)
//   __device__ bool  active(int m0, int n0)      : OPTIONAL - false: the tile at (m0, n0) is not computed at all (e.g. the
//                                                  lower tiles of a symmetric result, written by their mirror images)
//   __device__ void  select(int& tile, int& tiles_n) : OPTIONAL - called before setup(): picks one of several sub-problems
//                                                  from the flat tile index and rewrites it to that sub-problem's own
//   __device__ void  set_rowsq(const float* a, const float* b, int m0, int n0) : OPTIONAL (both operands K-contiguous) - the kernel sums the
//                                                  squares of the staged operand rows over all of K (a by-product of the staging
//                                                  pass) and hands the tile's TM + TM sums to the functor before the epilogue
//   __device__ void  epi4(int i0, int j, const float (&acc)[4], float* red) : OPTIONAL - four consecutive rows of one
//                                                  column at once (what one lane holds after the MFMA), all in range
//   static constexpr int A_NRAW / B_NRAW, raw_ok(), a_raw / a_fin, b_raw / b_fin : OPTIONAL - the two-phase operand path (see
//                                                  above gemm_group); a4 / b4 are then a_raw followed by a_fin
#pragma once
#include <type_traits>

#include "device_utils.h"

namespace adkf {

constexpr int GT = 64;        // default tile edge
constexpr int GTL = 128;      // large tile edge
constexpr int GK = 32;        // k chunk
constexpr int LD_MN = GK + 2; // [mn][k] layout, K-contiguous operands

typedef float f32x4 __attribute__((ext_vector_type(4)));

#ifndef ADKF_GEMM_ABLATE   // diagnostics (tools/gemm_bench.hip): 1 no fragment reads, 2 no barriers, 4 no operand loads after the first chunk, 8 no epilogue
#define ADKF_GEMM_ABLATE 0
#endif
#if (ADKF_GEMM_ABLATE & 2)
#define ADKF_GEMM_SYNC() __builtin_amdgcn_sched_barrier(0)
#else
#define ADKF_GEMM_SYNC() __syncthreads()
#endif

template <class P, class = void> struct has_skip : std::false_type {};
template <class P> struct has_skip<P, std::void_t<decltype(&P::skip)>> : std::true_type {};
template <class P, class = void> struct has_active : std::false_type {};
template <class P> struct has_active<P, std::void_t<decltype(&P::active)>> : std::true_type {};
template <class P, class = void> struct has_select : std::false_type {};
template <class P> struct has_select<P, std::void_t<decltype(&P::select)>> : std::true_type {};
template <class P, class = void> struct has_map : std::false_type {};      // the problem maps workgroups to (task, tile) itself (large.h: task groups)
template <class P> struct has_map<P, std::void_t<decltype(&P::map)>> : std::true_type {};
template <class P, class = void> struct has_epi4 : std::false_type {};
template <class P> struct has_epi4<P, std::void_t<decltype(&P::epi4)>> : std::true_type {};

template <class P, class = void> struct has_rowsq : std::false_type {};
template <class P> struct has_rowsq<P, std::void_t<decltype(&P::set_rowsq)>> : std::true_type {};
template <class P, class = void> struct has_rowsum : std::false_type {};
template <class P> struct has_rowsum<P, std::void_t<decltype(&P::set_rowsum)>> : std::true_type {};
// DEEP (with the two-phase operand path): when K is exactly DEEP chunks, the raw loads of ALL chunks are issued before the first
// MFMA - one trip to memory per workgroup instead of one per chunk.  For the K = 128 products of the blocked sweep (large.h), whose
// launches are a few workgroups per CU and therefore as long as ONE workgroup's chain of dependent loads.
template <class P, class = void> struct has_deep : std::false_type {};
template <class P> struct has_deep<P, std::void_t<decltype(P::DEEP)>> : std::true_type {};
// pre4 / epi4p: the epilogue's own operand (the matrix tile a product is subtracted from) is fetched with the operands, not after the last MFMA
template <class P, class = void> struct has_pre : std::false_type {};
template <class P> struct has_pre<P, std::void_t<decltype(&P::pre4)>> : std::true_type {};
template <class P, class = void> struct has_raw : std::false_type {};
template <class P> struct has_raw<P, std::void_t<decltype(P::A_NRAW)>> : std::true_type {};

template <int TM> struct GemmCfg {
    static constexpr int GPT = TM * GK / 256;   // operand elements staged per thread per chunk
    static constexpr int LD_K = TM + 16;        // [k][mn] layout, MN-contiguous operands (LD % 32 == 16)
    static constexpr int MNQ = TM / 4;          // float4 groups along mn
    static constexpr int KSTEP = 256 / MNQ;     // k rows covered per pass of the MN-contiguous map
    static constexpr int MI = TM / 32;          // 16 x 16 MFMA tiles per wave per dimension (2 x 2 waves)
};

// Each thread stages GPT operand entries per chunk as GPT / 4 groups of 4 that are consecutive along the operand's
// contiguous direction: K-contiguous -> (row r, k4..k4+3), MN-contiguous -> (k, mn4..mn4+3).
template <class P, bool IS_A, int TM>
__device__ __forceinline__ void gemm_fetch(const P& p, float (&reg)[GemmCfg<TM>::GPT], int base, int k0, int lim, int K) {
    using C = GemmCfg<TM>;
    constexpr bool KC = IS_A ? P::A_KCONTIG : P::B_KCONTIG;
    const int tid = threadIdx.x;
#pragma unroll
    for (int ps = 0; ps < C::GPT / 4; ++ps) {
        float v[4];
        if (KC) {
            const int g = base + (tid >> 3) + ps * 32, gk = k0 + (tid & 7) * 4;
            if (p.vec && g < lim && gk + 3 < K) {
                if (IS_A) p.a4(g, gk, v); else p.b4(gk, g, v);
            } else {
#pragma unroll
                for (int x = 0; x < 4; ++x) v[x] = (g < lim && gk + x < K) ? (IS_A ? p.a(g, gk + x) : p.b(gk + x, g)) : 0.f;
            }
        } else {
            const int g = base + (tid % C::MNQ) * 4, gk = k0 + (tid / C::MNQ) + ps * C::KSTEP;
            if (p.vec && g + 3 < lim && gk < K) {
                if (IS_A) p.a4(g, gk, v); else p.b4(gk, g, v);
            } else {
#pragma unroll
                for (int x = 0; x < 4; ++x) v[x] = (g + x < lim && gk < K) ? (IS_A ? p.a(g + x, gk) : p.b(gk, g + x)) : 0.f;
            }
        }
#pragma unroll
        for (int x = 0; x < 4; ++x) reg[ps * 4 + x] = v[x];
    }
}

// SQ: what rides the staging of a K-contiguous operand - 0 nothing, 1 the row sums of squares (ProbDist's norms), 2 the plain
// row sums (ProbDZ's coefficient vectors: coef_i = sum_k a(i, k), so no separate pass over the weight matrices)
template <bool KC, int TM, int SQ = 0>
__device__ __forceinline__ void gemm_stage(float* S, const float (&reg)[GemmCfg<TM>::GPT], float* sq = nullptr) {
    using C = GemmCfg<TM>;
    const int tid = threadIdx.x;
#pragma unroll
    for (int ps = 0; ps < C::GPT / 4; ++ps) {
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            if (KC) S[((tid >> 3) + ps * 32) * LD_MN + (tid & 7) * 4 + x] = reg[ps * 4 + x];
            else S[((tid / C::MNQ) + ps * C::KSTEP) * C::LD_K + (tid % C::MNQ) * 4 + x] = reg[ps * 4 + x];
            if constexpr (SQ == 1) sq[ps] = fmaf(reg[ps * 4 + x], reg[ps * 4 + x], sq[ps]);   // row (tid >> 3) + 32 ps, this thread's k group
            if constexpr (SQ == 2) sq[ps] += reg[ps * 4 + x];
        }
    }
}

// Two-phase operand path (problems with A_NRAW / B_NRAW): x_raw() only ISSUES the 16-byte loads of a group into registers,
// x_fin() turns them into the four operand entries (exp, scaling, centring ...).  The main loop issues the raw loads of
// chunk c+1 before the MFMAs of chunk c and runs x_fin() after them, while staging, so no wave waits for memory in front
// of its matrix instructions.  Used only for tiles that lie fully inside the task with K a multiple of the chunk (no
// per-lane range checks, hence no divergent branches around the loads); every other tile takes the checked path below.
template <class P, bool IS_A, int TM>
__device__ __forceinline__ void gemm_group(int ps, int base, int k0, int& g, int& gk) {
    using C = GemmCfg<TM>;
    constexpr bool KC = IS_A ? P::A_KCONTIG : P::B_KCONTIG;
    const int tid = threadIdx.x;
    if (KC) { g = base + (tid >> 3) + ps * 32; gk = k0 + (tid & 7) * 4; }
    else { g = base + (tid % C::MNQ) * 4; gk = k0 + (tid / C::MNQ) + ps * C::KSTEP; }
}

template <class P, bool IS_A, int TM, int NR>
__device__ __forceinline__ void gemm_fetch_raw(const P& p, float4 (&raw)[GemmCfg<TM>::GPT / 4][NR], int base, int k0) {
#pragma unroll
    for (int ps = 0; ps < GemmCfg<TM>::GPT / 4; ++ps) {
        int g, gk;
        gemm_group<P, IS_A, TM>(ps, base, k0, g, gk);
        if constexpr (IS_A) p.a_raw(g, gk, raw[ps]); else p.b_raw(gk, g, raw[ps]);
    }
}

template <class P, bool IS_A, int TM, int NR, int SQ = 0>
__device__ __forceinline__ void gemm_stage_raw(const P& p, float* S, const float4 (&raw)[GemmCfg<TM>::GPT / 4][NR], int base, int k0, float* sq = nullptr) {
    using C = GemmCfg<TM>;
    constexpr bool KC = IS_A ? P::A_KCONTIG : P::B_KCONTIG;
    const int tid = threadIdx.x;
#pragma unroll
    for (int ps = 0; ps < C::GPT / 4; ++ps) {
        int g, gk;
        gemm_group<P, IS_A, TM>(ps, base, k0, g, gk);
        float v[4];
        if constexpr (IS_A) p.a_fin(g, gk, raw[ps], v); else p.b_fin(gk, g, raw[ps], v);
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            if (KC) S[((tid >> 3) + ps * 32) * LD_MN + (tid & 7) * 4 + x] = v[x];
            else S[((tid / C::MNQ) + ps * C::KSTEP) * C::LD_K + (tid % C::MNQ) * 4 + x] = v[x];
            if constexpr (SQ == 1) sq[ps] = fmaf(v[x], v[x], sq[ps]);
            if constexpr (SQ == 2) sq[ps] += v[x];
        }
    }
}

template <class P, int TM = GT>
__global__ __launch_bounds__(256) void k_bgemm(P p, int T, int tiles_m, int tiles_n) {
    using C = GemmCfg<TM>;
    constexpr int MI = C::MI, LD_K = C::LD_K, GPT = C::GPT, WT = TM / 2;
    int task, tile;
    if constexpr (has_map<P>::value) {
        if (!p.map(tiles_m * tiles_n, task, tile)) return;
    } else {
        if (!task_tile(T, tiles_m * tiles_n, task, tile)) return;
    }
    if constexpr (has_select<P>::value) p.select(tile, tiles_n);   // several sub-problems in one launch (tiles_m = 1, tiles_n = all tiles)
    if (!p.setup(task)) return;
    const int M = p.M(), N = p.N();
    int K = p.K();
    const int m0 = (tile / tiles_n) * TM, n0 = (tile % tiles_n) * TM;
    if (m0 >= M || n0 >= N) {  // tile outside this (ragged) task: contributes zero partials
        if (P::NRED > 0 && threadIdx.x == 0) {
            float z[(P::NRED > 0 ? P::NRED : 1)];
            for (int q = 0; q < (P::NRED > 0 ? P::NRED : 1); ++q) z[q] = 0.f;
            p.store_red(tile, z);
        }
        return;
    }

    __shared__ float As[(P::A_KCONTIG ? TM * LD_MN : GK * LD_K)];
    __shared__ float Bs[(P::B_KCONTIG ? TM * LD_MN : GK * LD_K)];
    __shared__ float red_s[(P::NRED > 0 ? P::NRED * 4 : 1)];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 1, wc = wv & 1;  // 2x2 waves, WT x WT each
    const int fi = lane & 15, fk = lane >> 4;

    f32x4 acc[MI][MI];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    if constexpr (has_active<P>::value) {
        if (!p.active(m0, n0)) {   // not computed here: contributes zero partials
            if (P::NRED > 0 && threadIdx.x == 0) {
                float z[(P::NRED > 0 ? P::NRED : 1)];
                for (int q = 0; q < (P::NRED > 0 ? P::NRED : 1); ++q) z[q] = 0.f;
                p.store_red(tile, z);
            }
            return;
        }
    }
    if constexpr (has_skip<P>::value) {
        if (p.skip(m0, n0)) K = 0;
    }
    auto multiply_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < GK / 4; ++s) {
            float af[MI], bf[MI];
#if (ADKF_GEMM_ABLATE & 1)   // tools/gemm_bench.hip: no fragment reads
#pragma unroll
            for (int i = 0; i < MI; ++i) { af[i] = (float)(lane + s + i); bf[i] = (float)(lane - s - i); }
#else
#pragma unroll
            for (int i = 0; i < MI; ++i) {
                const int r = wr * WT + i * 16 + fi;
                af[i] = P::A_KCONTIG ? As[r * LD_MN + 4 * s + fk] : As[(4 * s + fk) * LD_K + r];
            }
#pragma unroll
            for (int j = 0; j < MI; ++j) {
                const int c = wc * WT + j * 16 + fi;
                bf[j] = P::B_KCONTIG ? Bs[c * LD_MN + 4 * s + fk] : Bs[(4 * s + fk) * LD_K + c];
            }
#endif
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };
    constexpr bool SQ = has_rowsq<P>::value;
    constexpr int SQA = has_rowsq<P>::value ? 1 : has_rowsum<P>::value ? 2 : 0, SQB = has_rowsq<P>::value ? 1 : 0;
    static_assert(!SQ || (P::A_KCONTIG && P::B_KCONTIG), "row sums of squares ride the K-contiguous staging map");
    static_assert(SQA != 2 || P::A_KCONTIG, "row sums ride the K-contiguous staging map");
    float sqa[GPT / 4], sqb[GPT / 4];
#pragma unroll
    for (int ps = 0; ps < GPT / 4; ++ps) { sqa[ps] = 0.f; sqb[ps] = 0.f; }
    bool fast = false, deep = false;
    float pre[has_pre<P>::value ? MI : 1][has_pre<P>::value ? MI : 1][4];
    if constexpr (has_deep<P>::value && has_raw<P>::value) {
        static_assert(has_pre<P>::value && has_epi4<P>::value, "a DEEP problem prefetches its epilogue operand");
        constexpr int DEEP = P::DEEP;
        deep = p.vec && m0 + TM <= M && n0 + TM <= N && K == DEEP * GK && p.raw_ok();
        if (deep) {
            float4 qa[DEEP][GPT / 4][P::A_NRAW], qb[DEEP][GPT / 4][P::B_NRAW];
#pragma unroll
            for (int c = 0; c < DEEP; ++c) {
                gemm_fetch_raw<P, true, TM, P::A_NRAW>(p, qa[c], m0, c * GK);
                gemm_fetch_raw<P, false, TM, P::B_NRAW>(p, qb[c], n0, c * GK);
            }
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < MI; ++j) p.pre4(m0 + wr * WT + i * 16 + fk * 4, n0 + wc * WT + j * 16 + fi, pre[i][j]);
#pragma unroll
            for (int c = 0; c < DEEP; ++c) {
                gemm_stage_raw<P, true, TM, P::A_NRAW, 0>(p, As, qa[c], m0, c * GK, sqa);
                gemm_stage_raw<P, false, TM, P::B_NRAW, 0>(p, Bs, qb[c], n0, c * GK, sqb);
                ADKF_GEMM_SYNC();
                multiply_chunk();
                ADKF_GEMM_SYNC();
            }
        }
    }
#ifndef ADKF_GEMM_NO_RAW
    if (deep) {} else   // diagnostics: -DADKF_GEMM_NO_RAW sends every tile through the checked path (tools/history/ab_lib.py)
    if constexpr (has_raw<P>::value) {
        fast = p.vec && m0 + TM <= M && n0 + TM <= N && K > 0 && (K % GK) == 0 && p.raw_ok();
        if (fast) {
            float4 qa[GPT / 4][P::A_NRAW], qb[GPT / 4][P::B_NRAW];
            gemm_fetch_raw<P, true, TM, P::A_NRAW>(p, qa, m0, 0);
            gemm_fetch_raw<P, false, TM, P::B_NRAW>(p, qb, n0, 0);
            for (int k0 = 0; k0 < K; k0 += GK) {
                gemm_stage_raw<P, true, TM, P::A_NRAW, SQA>(p, As, qa, m0, k0, sqa);
                gemm_stage_raw<P, false, TM, P::B_NRAW, SQB>(p, Bs, qb, n0, k0, sqb);
                ADKF_GEMM_SYNC();
                if (k0 + GK < K && !(ADKF_GEMM_ABLATE & 4)) {
                    gemm_fetch_raw<P, true, TM, P::A_NRAW>(p, qa, m0, k0 + GK);
                    gemm_fetch_raw<P, false, TM, P::B_NRAW>(p, qb, n0, k0 + GK);
                }
                __builtin_amdgcn_sched_barrier(0);   // the loads above are in flight before the first MFMA issues
                multiply_chunk();
                ADKF_GEMM_SYNC();
            }
        }
    }
#endif
    if (!fast && !deep) {
        float ra[GPT], rb[GPT];
        gemm_fetch<P, true, TM>(p, ra, m0, 0, M, K);
        gemm_fetch<P, false, TM>(p, rb, n0, 0, N, K);
        for (int k0 = 0; k0 < K; k0 += GK) {
            gemm_stage<P::A_KCONTIG, TM, SQA>(As, ra, sqa);
            gemm_stage<P::B_KCONTIG, TM, SQB>(Bs, rb, sqb);
            __syncthreads();
            if (k0 + GK < K) {  // next chunk's loads fly while this chunk is multiplied
                gemm_fetch<P, true, TM>(p, ra, m0, k0 + GK, M, K);
                gemm_fetch<P, false, TM>(p, rb, n0, k0 + GK, N, K);
            }
            multiply_chunk();
            __syncthreads();
        }
    }

    if constexpr (SQ) {
        // the eight threads that staged a row sit in eight adjacent lanes: three DPP steps, then one LDS hop to the epilogue's lanes
        __shared__ float rowsq[2][TM];
#pragma unroll
        for (int ps = 0; ps < GPT / 4; ++ps) {
            float a = sqa[ps], b = sqb[ps];
            a += dpp_f<DPP_XOR1>(a); a += dpp_f<DPP_XOR2>(a); a += dpp_f<DPP_HALF_MIRROR>(a);
            b += dpp_f<DPP_XOR1>(b); b += dpp_f<DPP_XOR2>(b); b += dpp_f<DPP_HALF_MIRROR>(b);
            if ((tid & 7) == 0) { rowsq[0][(tid >> 3) + ps * 32] = a; rowsq[1][(tid >> 3) + ps * 32] = b; }
        }
        __syncthreads();
        p.set_rowsq(&rowsq[0][0], &rowsq[1][0], m0, n0);
    }

    if constexpr (SQA == 2) {   // the plain row sums of A, same lane arithmetic
        __shared__ float rowsum[TM];
#pragma unroll
        for (int ps = 0; ps < GPT / 4; ++ps) {
            float a = sqa[ps];
            a += dpp_f<DPP_XOR1>(a); a += dpp_f<DPP_XOR2>(a); a += dpp_f<DPP_HALF_MIRROR>(a);
            if ((tid & 7) == 0) rowsum[(tid >> 3) + ps * 32] = a;
        }
        __syncthreads();
        p.set_rowsum(&rowsum[0], m0);
    }

    // ---- epilogue: C/D map of the 16x16 MFMA: col = lane & 15, row = (lane >> 4) * 4 + reg ----
    float red[(P::NRED > 0 ? P::NRED : 1)];
#pragma unroll
    for (int q = 0; q < (P::NRED > 0 ? P::NRED : 1); ++q) red[q] = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MI; ++j) {
            const int gi0 = m0 + wr * WT + i * 16 + fk * 4;
            const int gj = n0 + wc * WT + j * 16 + fi;
#if (ADKF_GEMM_ABLATE & 8)   // no epilogue (the accumulators stay alive through a store that never happens)
            if (acc[i][j][0] != 123.456f) continue;
#endif
            if constexpr (has_pre<P>::value) {
                if (deep) {   // (a deep tile lies fully inside the task)
                    const float v4[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    p.epi4p(gi0, gj, v4, pre[i][j], red);
                    continue;
                }
            }
            if constexpr (has_epi4<P>::value) {
                if (gi0 + 3 < M && gj < N) {
                    const float v4[4] = {acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]};
                    p.epi4(gi0, gj, v4, red);
                    continue;
                }
            }
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (gi0 + r < M && gj < N) p.epi(gi0 + r, gj, acc[i][j][r], red);
        }
    if (P::NRED > 0) {
        block_sum<(P::NRED > 0 ? P::NRED : 1), 256>(red, red_s);
        if (tid == 0) p.store_red(tile, red);
    }
}

}  // namespace adkf
