// k_bgemm3: gemm.h's batched GEMM with the FP32 products formed on the BF16 matrix pipe.
//
// MI355X runs v_mfma_f32_16x16x4_f32 at the FP32 vector rate (157 TFLOP/s), a sixteenth of the BF16 rate.  A float is the EXACT sum of
// three bfloat16 values (its 24 significant bits cut into 8 + 8 + 8: x0 = bf16(x), x1 = bf16(x - x0), x2 = x - x0 - x1, each
// cut rounding to nearest; same exponent range, so no scaling), and a product of two bfloat16 values is exact in the FP32 accumulator, hence
//     x y = x0 y0 + (x0 y1 + x1 y0) + (x0 y2 + x1 y1 + x2 y0) + [x1 y2 + x2 y1 + x2 y2: below 2^-24 |x y|, dropped]
// is an FP32 product to FP32 accuracy out of SIX v_mfma_f32_16x16x32_bf16 (16 cycles each, K = 32) where the FP32 form takes eight
// v_mfma_f32_16x16x4_f32 (32 cycles each): 96 against 256 matrix-pipe cycles per 16 x 16 x 32 block.  The error of a whole dot product
// (tests/test_x3_split.py, emulated in numpy: 3e-9 of sum |x||y| for the split itself, 1e-7 with FP32 accumulation) is that of an FP32
// FMA chain; nothing is rounded to bfloat16 anywhere.  The price is the splitting - eleven VALU instructions per pair of operand
// entries, paid once per staged entry - so the form pays where a staged entry is used by 64 or more MFMA columns: the N^2 d products
// (squared distances, dL/dZ) and the N^3 products of the > 128-point path, not the per-task kernels whose operands are re-read from LDS
// by every wave.
//
// Accumulation (measured, tools/x3_bench.hip: mean signed error of the inner products): the matrix pipe's add of SMALL products into a
// LARGE accumulator loses them one-sidedly (all six terms in one accumulator: inner products biased by 1e-8 relative, ten times the
// FP32 form - which a batch of correlated kernel-matrix entries does not average out).  So the leading term x0 y0 has the
// accumulator to itself and the five small terms share a second one, added once at the end: bias and maximum error are then at or
// below those of the FP32 form (-1e-9 / 1.7e-7 against -9e-10 / 1.8e-7 at d = 2; 2e-9 / 4e-7 against 2e-9 / 1e-6 at d = 256).
//
// Same problem-functor interface, K chunk (32 = one MFMA step), epilogue and C/D register map (dtype-independent on gfx950) as
// k_bgemm; TM x TM output tile per workgroup of NT lanes: 64 x 64 with 256 (2 x 2 waves of 32 x 32) or 128 x 128 with 512 (2 x 4 waves
// of 64 x 32: half the splitting work per MFMA, two waves per SIMD).  LDS image per operand: three planes [TM][32] of bfloat16,
// K-contiguous whatever the memory layout, rows 80 bytes apart (a lane's fragment is ONE ds_read_b128 - row l & 15, k = 8 (l >> 4) ..
// + 7 - and the sixteen rows of a group of lanes fall on all 64 banks).  K-contiguous operands are staged by the thread -> element map
// of gemm.h (row tid / 8 + 32 ps, four k per thread); MN-contiguous ones (dL/dZ's feature matrix: [k][column]) with a COLUMN per lane -
// GPT consecutive k of it from GPT scalar loads, each a coalesced row segment across the wave - so that the transposition is free:
// the lane holds exactly the k-run its LDS row wants and writes it as one 16-byte store per plane, on the conflict-free row stride.
// (First version: four columns x two k per thread from 16-byte loads, twelve 4-byte LDS stores with four-way bank conflicts -
// dL/dZ came out 5 % SLOWER than on the FP32 pipe.)
#pragma once
#include "gemm.h"

namespace adkf {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int X3_RS = 40;   // LDS row stride in bfloat16 (80 bytes)
#ifndef ADKF_X3_ABLATE   // diagnostics (tools/x3_bench.hip): 1 no MFMAs, 2 no splitting arithmetic, 4 no operand loads after the first chunk, 8 no epilogue, 16 no LDS fragment reads
#define ADKF_X3_ABLATE 0
#endif

// two floats -> their three bfloat16 pieces, packed pairwise (low half: a, high half: b).  Round to nearest at every cut (v_cvt_pk_bf16_f32):
// the pieces still sum to the float exactly (the remainders have 16, then 8 significant bits), and - unlike cutting by truncation,
// which was tried first - the remainders carry random signs, so the three dropped terms average out instead of biasing every inner
// product low by 2^-25 (measured: tools/x3_bench.hip).
__device__ __forceinline__ uint32_t x3_pack(float a, float b) {
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const f32x2 v = {a, b};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ void x3_split2(float a, float b, uint32_t& p0, uint32_t& p1, uint32_t& p2) {
#if (ADKF_X3_ABLATE & 2)
    p0 = __float_as_uint(a); p1 = __float_as_uint(b); p2 = p0 ^ p1; return;
#endif
    p0 = x3_pack(a, b);
    const float ra = a - __uint_as_float(p0 << 16), rb = b - __uint_as_float(p0 & 0xffff0000u);   // exact
    p1 = x3_pack(ra, rb);
    const float sa = ra - __uint_as_float(p1 << 16), sb = rb - __uint_as_float(p1 & 0xffff0000u);   // exact; fits bfloat16
    p2 = x3_pack(sa, sb);
}

template <int TM, int NT> struct X3Cfg {
    static_assert((TM == 64 && NT == 256) || (TM == 128 && NT == 512) || (TM == 128 && NT == 256), "tile / workgroup shapes of k_bgemm3");
    static constexpr int GPT = TM * GK / NT;     // operand entries per thread and chunk (8 or 16)
    static constexpr int NG = GPT / 4;           // groups of four per thread
    static constexpr int ROWS = NT / 8;          // K-contiguous map: rows per pass (eight threads per row)
    static constexpr int KQ = NT / TM;           // MN-contiguous map: lane -> column tid % TM, k run (tid / TM) * GPT .. + GPT - 1
    static_assert(KQ * GPT == GK && (GPT % 8) == 0, "a thread's k run is one or two 16-byte LDS stores");
    static constexpr int PLANE = TM * X3_RS;     // bfloat16 per plane
    static constexpr int WC = NT / 128, WR = 2;  // waves: WR x WC, each (TM / WR) x (TM / WC)
    static constexpr int MI = TM / WR / 16, MJ = TM / WC / 16;
};

// K-contiguous map: (row g, first k gk) of group ps of this thread
template <int TM, int NT>
__device__ __forceinline__ void x3_group(int ps, int base, int k0, int& g, int& gk) {
    using C = X3Cfg<TM, NT>;
    const int tid = threadIdx.x;
    g = base + (tid >> 3) + ps * C::ROWS; gk = k0 + (tid & 7) * 4;
}

// K-contiguous: v[ps][x] = entry (row of group ps, k4 + x).  MN-contiguous: v[h][x] = entry (k run position 4 h + x, this lane's column).
template <bool KC, int TM, int NT, int SQ>
__device__ __forceinline__ void x3_stage(unsigned short* S, const float (&v)[X3Cfg<TM, NT>::NG][4], float* sq) {
    using C = X3Cfg<TM, NT>;
    const int tid = threadIdx.x;
    if constexpr (KC) {
#pragma unroll
        for (int ps = 0; ps < C::NG; ++ps) {
            uint32_t a0, a1, a2, b0, b1, b2;
            x3_split2(v[ps][0], v[ps][1], a0, a1, a2);
            x3_split2(v[ps][2], v[ps][3], b0, b1, b2);
            unsigned short* d = S + ((tid >> 3) + ps * C::ROWS) * X3_RS + (tid & 7) * 4;
            *reinterpret_cast<uint2*>(d) = make_uint2(a0, b0);
            *reinterpret_cast<uint2*>(d + C::PLANE) = make_uint2(a1, b1);
            *reinterpret_cast<uint2*>(d + 2 * C::PLANE) = make_uint2(a2, b2);
#pragma unroll
            for (int x = 0; x < 4; ++x) {
                if constexpr (SQ == 1) sq[ps] = fmaf(v[ps][x], v[ps][x], sq[ps]);
                if constexpr (SQ == 2) sq[ps] += v[ps][x];
            }
        }
    } else {
        static_assert(SQ == 0, "row sums ride the K-contiguous staging map");
        unsigned short* d = S + (tid % TM) * X3_RS + (tid / TM) * C::GPT;
#pragma unroll
        for (int h = 0; h < C::NG; h += 2) {   // eight consecutive k: one 16-byte store per plane
            uint32_t p0[4], p1[4], p2[4];
            x3_split2(v[h][0], v[h][1], p0[0], p1[0], p2[0]);
            x3_split2(v[h][2], v[h][3], p0[1], p1[1], p2[1]);
            x3_split2(v[h + 1][0], v[h + 1][1], p0[2], p1[2], p2[2]);
            x3_split2(v[h + 1][2], v[h + 1][3], p0[3], p1[3], p2[3]);
            *reinterpret_cast<uint4*>(d + 4 * h) = make_uint4(p0[0], p0[1], p0[2], p0[3]);
            *reinterpret_cast<uint4*>(d + 4 * h + C::PLANE) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
            *reinterpret_cast<uint4*>(d + 4 * h + 2 * C::PLANE) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
        }
    }
}

// MN-contiguous operand: this lane's column, its GPT consecutive k, by the functor's scalar accessor (CHECKED: range-checked)
template <class P, bool IS_A, int TM, int NT, bool CHECKED>
__device__ __forceinline__ void x3_fetch_col(const P& p, float (&v)[X3Cfg<TM, NT>::NG][4], int base, int k0, int lim, int K) {
    using C = X3Cfg<TM, NT>;
    const int g = base + (threadIdx.x % TM), kb = k0 + (threadIdx.x / TM) * C::GPT;
#pragma unroll
    for (int h = 0; h < C::NG; ++h)
#pragma unroll
        for (int x = 0; x < 4; ++x) {
            const int k = kb + 4 * h + x;
            if (CHECKED) v[h][x] = (g < lim && k < K) ? (IS_A ? p.a(g, k) : p.b(k, g)) : 0.f;
            else v[h][x] = IS_A ? p.a(g, k) : p.b(k, g);
        }
}

template <class P, bool IS_A, int TM, int NT, int NR>
__device__ __forceinline__ void x3_fetch_raw(const P& p, float4 (&raw)[X3Cfg<TM, NT>::NG][NR], int base, int k0) {
#pragma unroll
    for (int ps = 0; ps < X3Cfg<TM, NT>::NG; ++ps) {
        int g, gk;
        x3_group<TM, NT>(ps, base, k0, g, gk);
        if constexpr (IS_A) p.a_raw(g, gk, raw[ps]); else p.b_raw(gk, g, raw[ps]);
    }
}

template <class P, bool IS_A, int TM, int NT, int NR, int SQ>
__device__ __forceinline__ void x3_stage_raw(const P& p, unsigned short* S, const float4 (&raw)[X3Cfg<TM, NT>::NG][NR], int base, int k0, float* sq) {
    float v[X3Cfg<TM, NT>::NG][4];
#pragma unroll
    for (int ps = 0; ps < X3Cfg<TM, NT>::NG; ++ps) {
        int g, gk;
        x3_group<TM, NT>(ps, base, k0, g, gk);
        if constexpr (IS_A) p.a_fin(g, gk, raw[ps], v[ps]); else p.b_fin(gk, g, raw[ps], v[ps]);
    }
    x3_stage<true, TM, NT, SQ>(S, v, sq);
}

// range-checked fetch of a K-contiguous operand (ragged tiles, K not a multiple of the chunk, problems without the two-phase path)
template <class P, bool IS_A, int TM, int NT>
__device__ __forceinline__ void x3_fetch(const P& p, float (&v)[X3Cfg<TM, NT>::NG][4], int base, int k0, int lim, int K) {
    constexpr bool KC = IS_A ? P::A_KCONTIG : P::B_KCONTIG;
    if constexpr (!KC) { x3_fetch_col<P, IS_A, TM, NT, true>(p, v, base, k0, lim, K); return; }
#pragma unroll
    for (int ps = 0; ps < X3Cfg<TM, NT>::NG; ++ps) {
        int g, gk;
        x3_group<TM, NT>(ps, base, k0, g, gk);
        if (p.vec && g < lim && gk + 3 < K) {
            if (IS_A) p.a4(g, gk, v[ps]); else p.b4(gk, g, v[ps]);
        } else {
#pragma unroll
            for (int x = 0; x < 4; ++x) v[ps][x] = (g < lim && gk + x < K) ? (IS_A ? p.a(g, gk + x) : p.b(gk + x, g)) : 0.f;
        }
    }
}

template <class P, int TM = GT, int NT = 256>
__global__ __launch_bounds__(NT) void k_bgemm3(P p, int T, int tiles_m, int tiles_n) {
    using C = X3Cfg<TM, NT>;
    constexpr int MI = C::MI, MJ = C::MJ, NG = C::NG, WTR = TM / C::WR, WTC = TM / C::WC;
    int task, tile;
    if constexpr (has_map<P>::value) {
        if (!p.map(tiles_m * tiles_n, task, tile)) return;
    } else {
        if (!task_tile(T, tiles_m * tiles_n, task, tile)) return;
    }
    if constexpr (has_select<P>::value) p.select(tile, tiles_n);
    if (!p.setup(task)) return;
    const int M = p.M(), N = p.N();
    int K = p.K();
    const int m0 = (tile / tiles_n) * TM, n0 = (tile % tiles_n) * TM;
    bool nothing = m0 >= M || n0 >= N;
    if constexpr (has_active<P>::value) {
        if (!nothing) nothing = !p.active(m0, n0);
    }
    if (nothing) {  // tile outside this (ragged) task, or not computed here: contributes zero partials
        if (P::NRED > 0 && threadIdx.x == 0) {
            float z[(P::NRED > 0 ? P::NRED : 1)];
            for (int q = 0; q < (P::NRED > 0 ? P::NRED : 1); ++q) z[q] = 0.f;
            p.store_red(tile, z);
        }
        return;
    }

    __shared__ __attribute__((aligned(16))) unsigned short As[3 * C::PLANE];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[3 * C::PLANE];
    __shared__ __attribute__((aligned(16))) float red_s[(P::NRED > 0 ? P::NRED * (NT / 64) : 1)];

    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv / C::WC, wc = wv % C::WC;
    const int fi = lane & 15, fk = lane >> 4;

    f32x4 acc[MI][MJ], small[MI][MJ];   // the leading terms x0 y0; the five small ones (header: why they are kept apart)
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) { acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; small[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    if constexpr (has_skip<P>::value) {
        if (p.skip(m0, n0)) K = 0;
    }
    auto multiply_chunk = [&]() __attribute__((always_inline)) {
#if (ADKF_X3_ABLATE & 1)
        return;
#endif
        bf16x8 af[MI][3], bf[MJ][3];
#if (ADKF_X3_ABLATE & 16)
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q) af[i][q] = __builtin_bit_cast(bf16x8, make_uint4(lane + i, lane + q, i, q));
#pragma unroll
        for (int j = 0; j < MJ; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q) bf[j][q] = __builtin_bit_cast(bf16x8, make_uint4(lane + j, lane - q, j, q));
#else
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                af[i][q] = *reinterpret_cast<const bf16x8*>(As + q * C::PLANE + (wr * WTR + i * 16 + fi) * X3_RS + 8 * fk);
#pragma unroll
        for (int j = 0; j < MJ; ++j)
#pragma unroll
            for (int q = 0; q < 3; ++q)
                bf[j][q] = *reinterpret_cast<const bf16x8*>(Bs + q * C::PLANE + (wc * WTC + j * 16 + fi) * X3_RS + 8 * fk);
#endif
        // each term runs over all tiles, so that consecutive MFMAs never share an accumulator; smallest first
#define ADKF_X3_TERM(dst_, qa_, qb_)                                                                                            \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < MJ; ++j)                             \
        dst_[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i][qa_], bf[j][qb_], dst_[i][j], 0, 0, 0);
        ADKF_X3_TERM(small, 0, 2) ADKF_X3_TERM(small, 2, 0) ADKF_X3_TERM(small, 1, 1) ADKF_X3_TERM(acc, 0, 0)
        ADKF_X3_TERM(small, 0, 1) ADKF_X3_TERM(small, 1, 0)
#undef ADKF_X3_TERM
    };
    constexpr bool SQ = has_rowsq<P>::value;
    constexpr int SQA = has_rowsq<P>::value ? 1 : has_rowsum<P>::value ? 2 : 0, SQB = has_rowsq<P>::value ? 1 : 0;
    static_assert(!SQ || (P::A_KCONTIG && P::B_KCONTIG), "row sums of squares ride the K-contiguous staging map");
    static_assert(SQA != 2 || P::A_KCONTIG, "row sums ride the K-contiguous staging map");
    float sqa[NG], sqb[NG];
#pragma unroll
    for (int ps = 0; ps < NG; ++ps) { sqa[ps] = 0.f; sqb[ps] = 0.f; }
    bool fast = false;
    if constexpr (has_raw<P>::value) {
        fast = p.vec && m0 + TM <= M && n0 + TM <= N && K > 0 && (K % GK) == 0 && p.raw_ok();
        if (fast) {
            // K-contiguous operands: the two-phase path of gemm.h (16-byte loads issued a chunk ahead, functor arithmetic while staging);
            // MN-contiguous ones: the lane's k run by scalar loads, likewise a chunk ahead
            float4 qa[P::A_KCONTIG ? NG : 1][P::A_NRAW], qb[P::B_KCONTIG ? NG : 1][P::B_NRAW];
            float ca[P::A_KCONTIG ? 1 : NG][4], cb[P::B_KCONTIG ? 1 : NG][4];
            auto fetch = [&](int k0) __attribute__((always_inline)) {
                if constexpr (P::A_KCONTIG) x3_fetch_raw<P, true, TM, NT, P::A_NRAW>(p, qa, m0, k0);
                else x3_fetch_col<P, true, TM, NT, false>(p, ca, m0, k0, M, K);
                if constexpr (P::B_KCONTIG) x3_fetch_raw<P, false, TM, NT, P::B_NRAW>(p, qb, n0, k0);
                else x3_fetch_col<P, false, TM, NT, false>(p, cb, n0, k0, N, K);
            };
            fetch(0);
            for (int k0 = 0; k0 < K; k0 += GK) {
                if constexpr (P::A_KCONTIG) x3_stage_raw<P, true, TM, NT, P::A_NRAW, SQA>(p, As, qa, m0, k0, sqa);
                else x3_stage<false, TM, NT, 0>(As, ca, sqa);
                if constexpr (P::B_KCONTIG) x3_stage_raw<P, false, TM, NT, P::B_NRAW, SQB>(p, Bs, qb, n0, k0, sqb);
                else x3_stage<false, TM, NT, 0>(Bs, cb, sqb);
                __syncthreads();
                if (k0 + GK < K && !(ADKF_X3_ABLATE & 4)) fetch(k0 + GK);
                __builtin_amdgcn_sched_barrier(0);   // the loads above are in flight before the first MFMA issues
                multiply_chunk();
                __syncthreads();
            }
        }
    }
    if (!fast) {
        float ra[NG][4], rb[NG][4];
        x3_fetch<P, true, TM, NT>(p, ra, m0, 0, M, K);
        x3_fetch<P, false, TM, NT>(p, rb, n0, 0, N, K);
        for (int k0 = 0; k0 < K; k0 += GK) {
            x3_stage<P::A_KCONTIG, TM, NT, SQA>(As, ra, sqa);
            x3_stage<P::B_KCONTIG, TM, NT, SQB>(Bs, rb, sqb);
            __syncthreads();
            if (k0 + GK < K) {
                x3_fetch<P, true, TM, NT>(p, ra, m0, k0 + GK, M, K);
                x3_fetch<P, false, TM, NT>(p, rb, n0, k0 + GK, N, K);
            }
            multiply_chunk();
            __syncthreads();
        }
    }
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) acc[i][j] += small[i][j];

    if constexpr (SQ) {
        // the eight threads that staged a row sit in eight adjacent lanes: three DPP steps, then one LDS hop to the epilogue's lanes
        __shared__ float rowsq[2][TM];
#pragma unroll
        for (int ps = 0; ps < NG; ++ps) {
            float a = sqa[ps], b = sqb[ps];
            a += dpp_f<DPP_XOR1>(a); a += dpp_f<DPP_XOR2>(a); a += dpp_f<DPP_HALF_MIRROR>(a);
            b += dpp_f<DPP_XOR1>(b); b += dpp_f<DPP_XOR2>(b); b += dpp_f<DPP_HALF_MIRROR>(b);
            if ((tid & 7) == 0) { rowsq[0][(tid >> 3) + ps * C::ROWS] = a; rowsq[1][(tid >> 3) + ps * C::ROWS] = b; }
        }
        __syncthreads();
        p.set_rowsq(&rowsq[0][0], &rowsq[1][0], m0, n0);
    }
    if constexpr (SQA == 2) {   // the plain row sums of A, same lane arithmetic
        __shared__ float rowsum[TM];
#pragma unroll
        for (int ps = 0; ps < NG; ++ps) {
            float a = sqa[ps];
            a += dpp_f<DPP_XOR1>(a); a += dpp_f<DPP_XOR2>(a); a += dpp_f<DPP_HALF_MIRROR>(a);
            if ((tid & 7) == 0) rowsum[(tid >> 3) + ps * C::ROWS] = a;
        }
        __syncthreads();
        p.set_rowsum(&rowsum[0], m0);
    }

    // ---- epilogue: C/D map of the 16 x 16 MFMA (the same for every input type): col = lane & 15, row = (lane >> 4) * 4 + reg ----
    float red[(P::NRED > 0 ? P::NRED : 1)];
#pragma unroll
    for (int q = 0; q < (P::NRED > 0 ? P::NRED : 1); ++q) red[q] = 0.f;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            const int gi0 = m0 + wr * WTR + i * 16 + fk * 4;
            const int gj = n0 + wc * WTC + j * 16 + fi;
#if (ADKF_X3_ABLATE & 8)
            if (acc[i][j][0] != 123.456f) continue;
#endif
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
        block_sum<(P::NRED > 0 ? P::NRED : 1), NT>(red, red_s);
        if (tid == 0) p.store_red(tile, red);
    }
}

}  // namespace adkf
