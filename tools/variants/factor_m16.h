// Sweep<128, 512> in SIXTEEN-pivot block steps, everything on the matrix pipe.  EXPERIMENT, kept for A/B runs only
// (-DADKF_SWEEP_M=2; tools/sweepm_bench.hip, tools/history/run_sweepm16.sh): correct for every n, but 43.0 k cycles per sweep against the
// 37.0 k of factor_m.h, which stays the default.  Measured (round 3, cycles per block step of 16 pivots, s_memtime):
//   * the chain alone, no bulk update at all: 3 400 (27.2 k per sweep) = barrier -> LDS reads -> -F (2 + 2 MFMAs) -> critical tile
//     (2 + 2 MFMAs) -> piece stored ~1 000, then four in-wave sub-steps of ~630 (gj4 220, the turn of D^-1 and the pivot rows
//     through LDS ~250 - a wave's own store -> read costs nearly what a cross-wave hand-off does, the barrier being only 44 of
//     it -, operand arithmetic + MFMA + its result ~160), inverse stored + barrier ~200;
//   * with the bulk update (28 MFMAs per wave and step) the chain wave's sub-steps take 900 - 1 900: every MFMA it issues
//     itself, also from inside the Gauss-Jordan stream (gj4m), costs it a matrix-pipe slot of 32 cycles, and its SIMD partner
//     saturates that pipe for the first ~1 000 cycles of the step; the other waves finish at 2 200 - 3 100 and wait.
// So the idea below - fewer, longer hand-offs - does not pay on this machine: what a step costs is the latency of moving
// sixteen numbers (D^-1) from one quad to the other lanes, and that is the same ~250 - 350 cycles inside a wave as between waves.
//
// factor_m.h hands over after every four pivots: barrier -> LDS reads -> A operand -> one MFMA -> piece / 4 x 4 inverse -> LDS
// store -> barrier, ~1150 cycles, 32 times per sweep, with the matrix pipe a third busy.  With the block order WAVE-major (the
// four blocks of one wave's 16 rows in a row) the 16 x 16 diagonal
// block E of a wave is swept inside that wave's own diagonal tile - four 4 x 4 Gauss-Jordan steps (gj4) joined by one MFMA
// each, operands turned through a few hundred bytes of LDS that only this wave touches, no barrier - and everybody else needs
// -E^-1 and the sixteen pivot rows only once per sixteen pivots.  Eight hand-offs per sweep instead of thirty-two.
//
// Layout: as in factor_m.h.  Wave w (0..7), lane l: p = l & 15, g = l >> 4; tile x of the wave's eight 16 x 16 accumulator
// tiles holds, in register y,  M[I = 16 w + 4 g + y][J = 16 x + p].
//
// Block step S (owner: wave S, pivots W = 16 S .. 16 S + 15), with factor.h's trick at block size 16 (C' = the pivot rows with
// E - I at the pivot columns, so ONE uniform update  M_ij -= sum_q F_q,i C'_q,j,  F = E^-1 C',  covers rows, columns and rest):
//   * what is in LDS when its barrier opens (slot S & 1): ct[column j][pivot q] = C'[q][j] - every wave delivered its own 16
//     columns from the TRANSPOSED tile it holds (tile S: its rows x the pivot columns), nobody brought a 128-wide row up to
//     date - and einv[m][k] = -E^-1, from the owner;
//   * every wave forms -F for its own 16 columns with four MFMAs (A = -E^-1, B = its columns of C'); the result,
//     lane (p, g) register y = -F[4 g + y][16 w + p], IS the A operand of its eight tile updates when the sixteen pivots are
//     taken in the order k-chunk y = {y, 4 + y, 8 + y, 12 + y}: no transposition, no LDS.  The B operand of tile x for all four
//     chunks is ONE 16-byte read, ct[16 x + p][4 g .. 4 g + 3];
//   * tile S + 1 first: it yields the wave's piece of the NEXT pivot rows and, in wave S + 1, the next E, which that wave then
//     sweeps in place (sub_chain) with its own 28 bulk MFMAs riding in the Gauss-Jordan stream and in the LDS waits; its result
//     -E^-1 stays in the tile (M_PP of the classical sweep), is published, and the tile is left out of that wave's next update.
#pragma once
#ifndef ADKF_M16_TURN
#define ADKF_M16_TURN 0   // (0 is faster: 26.5 k against 31.5 k cycles for the chain alone) 1: D^-1 reaches the other lanes as sixteen v_readlane -> SGPR operands of the A operand's FMAs, the pivot rows' read is issued in front of the Gauss-Jordan stream; 0: both through LDS behind it
#endif
#ifndef ADKF_M16_PACE
#define ADKF_M16_PACE 1   // s_sleep argument between pairs of the chain wave's SIMD partner's bulk MFMAs (0: none)
#endif
#ifndef ADKF_M_ABLATE
#define ADKF_M_ABLATE 0   // timing-only ablations for tools/sweepm_bench.hip (2: no 16 x 16 sweep, 8: no bulk MFMAs, 32: none of the chain wave's own)
#endif

namespace adkf {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <> struct SweepSmem<128, 512> {
    static constexpr int NSLOT = 2;
#ifndef ADKF_M16_CS
#define ADKF_M16_CS 20
#endif
    static constexpr int CS = ADKF_M16_CS;     // column stride in floats (20: conflict-free piece stores, two-way 16-byte reads; 24: the other way round)
    alignas(16) float ct[NSLOT][128][CS];      // C'^T: [column][pivot of the block step]
    alignas(16) float einv[NSLOT][16][CS];     // -E^-1
    alignas(16) float cw[16][4];               // the owner's in-wave turn-table: the four pivot rows of a sub-step, [column][k]
    alignas(16) float dw[4][4];                // ... and D^-1
    alignas(16) float pivs[128];
    alignas(16) float vec_in[128];
    alignas(16) float vec_out[128];
    float red[8 * 8];
    int redi[8];
#if ADKF_STAMP
    unsigned long long stamp[8 * 16];
#endif
    static constexpr int SCRATCH_FLOATS = NSLOT * 128 * CS;
    __device__ __forceinline__ float* scratch() { return &ct[0][0][0]; }   // free for the caller between two sweeps
};

#if ADKF_STAMP
#ifndef ADKF_STAMP_SITE
#define ADKF_STAMP_SITE 1
#endif
#define ADKF_MTS(slot_) do { if ((slot_ == 0 || slot_ == ADKF_STAMP_SITE || ADKF_STAMP_SITE < 0) && S == ADKF_STAMP && (threadIdx.x & 63) == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); sm.stamp[(threadIdx.x >> 6) * 16 + (slot_)] = t_; } } while (0)
#else
#define ADKF_MTS(slot_) do {} while (0)
#endif

template <> struct Sweep<128, 512> {
    using Smem = SweepSmem<128, 512>;
    static constexpr int NMAX = 128, NT = 512, RB = 8, CB = 4, B = 4, NW = 8, CS = Smem::CS;

    __device__ static __forceinline__ int wave() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }
    __device__ static __forceinline__ int bc() { return threadIdx.x >> 4; }   // column block: 4 w + g
    __device__ static __forceinline__ int row(int r) { return (r << 4) + (threadIdx.x & 15); }
    __device__ static __forceinline__ int col(int c) { return bc() * CB + c; }

    // In-place inverse of the 4 x 4 block D inside every quad (quad lane a holds row a in d[0..3]) by four Gauss-Jordan steps,
    // and its successive pivots (uniform in the quad): factor_m.h's hand-scheduled stream, unchanged (see there).
#define ADKF_M16_GJ_STEP(P, dP, dA, dB, dC, pv, mk, MF) \
        "v_mov_b32_dpp " pv ", " dP " quad_perm:[" #P "," #P "," #P "," #P "] row_mask:0xf bank_mask:0xf\n\t" \
        "v_rcp_f32_e32 %[r], " pv "\n\t" \
        MF \
        "v_cndmask_b32_e64 %[y], " dP ", -1.0, " mk "\n\t" \
        "v_cndmask_b32_e64 %[z], 0, -1.0, " mk "\n\t" \
        "v_fma_f32 %[g], -%[r], %[y], %[z]\n\t" \
        "v_fmac_f32_dpp " dA ", " dA ", %[g] quad_perm:[" #P "," #P "," #P "," #P "] row_mask:0xf bank_mask:0xf\n\t" \
        "v_fmac_f32_dpp " dB ", " dB ", %[g] quad_perm:[" #P "," #P "," #P "," #P "] row_mask:0xf bank_mask:0xf\n\t" \
        "v_fmac_f32_dpp " dC ", " dC ", %[g] quad_perm:[" #P "," #P "," #P "," #P "] row_mask:0xf bank_mask:0xf\n\t" \
        "v_cndmask_b32_e64 " dP ", %[g], %[r], " mk "\n\t"
#define ADKF_M16_GJ_BODY(MF0, MF1, MF2, MF3, TAIL) \
                     "s_nop 1\n\t" \
                     ADKF_M16_GJ_STEP(0, "%[d0]", "%[d1]", "%[d2]", "%[d3]", "%[p0]", "%[m0]", MF0) \
                     ADKF_M16_GJ_STEP(1, "%[d1]", "%[d2]", "%[d3]", "%[d0]", "%[p1]", "%[m1]", MF1) \
                     ADKF_M16_GJ_STEP(2, "%[d2]", "%[d3]", "%[d0]", "%[d1]", "%[p2]", "%[m2]", MF2) \
                     ADKF_M16_GJ_STEP(3, "%[d3]", "%[d0]", "%[d1]", "%[d2]", "%[p3]", "%[m3]", MF3) TAIL
    __device__ static __forceinline__ void gj4(float (&D)[4], float (&piv)[4]) {
        float r, y, z, g;
        const unsigned long long m0 = 0x1111111111111111ull, m1 = 0x2222222222222222ull, m2 = 0x4444444444444444ull, m3 = 0x8888888888888888ull;
        asm volatile(ADKF_M16_GJ_BODY("", "", "", "", "")
                     : [d0] "+v"(D[0]), [d1] "+v"(D[1]), [d2] "+v"(D[2]), [d3] "+v"(D[3]),
                       [p0] "=&v"(piv[0]), [p1] "=&v"(piv[1]), [p2] "=&v"(piv[2]), [p3] "=&v"(piv[3]),
                       [r] "=&v"(r), [y] "=&v"(y), [z] "=&v"(z), [g] "=&v"(g)
                     : [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2), [m3] "s"(m3));
    }
    // The same stream with the four MFMAs of one tile's update (acc += sum_y nf[y] b[y]) riding in it, one per pivot, behind the
    // reciprocal: the chain wave's own bulk update costs it four issue cycles per MFMA there instead of a matrix-pipe slot of
    // thirty-two at the end of its chain.  (s_nop at the end: an MFMA's result must not be read by whatever the compiler places
    // next for 12 wait states; nine instructions of the last pivot follow the last one.)
    __device__ static __forceinline__ void gj4m(float (&D)[4], float (&piv)[4], f32x4_t& acc, const f32x4_t& nf, const float4& b) {
        float r, y, z, g;
        const unsigned long long m0 = 0x1111111111111111ull, m1 = 0x2222222222222222ull, m2 = 0x4444444444444444ull, m3 = 0x8888888888888888ull;
        asm volatile(ADKF_M16_GJ_BODY("v_mfma_f32_16x16x4_f32 %[acc], %[n0], %[b0], %[acc]\n\t", "v_mfma_f32_16x16x4_f32 %[acc], %[n1], %[b1], %[acc]\n\t",
                                      "v_mfma_f32_16x16x4_f32 %[acc], %[n2], %[b2], %[acc]\n\t", "v_mfma_f32_16x16x4_f32 %[acc], %[n3], %[b3], %[acc]\n\t", "s_nop 2\n\t")
                     : [d0] "+v"(D[0]), [d1] "+v"(D[1]), [d2] "+v"(D[2]), [d3] "+v"(D[3]),
                       [p0] "=&v"(piv[0]), [p1] "=&v"(piv[1]), [p2] "=&v"(piv[2]), [p3] "=&v"(piv[3]),
                       [r] "=&v"(r), [y] "=&v"(y), [z] "=&v"(z), [g] "=&v"(g), [acc] "+v"(acc)
                     : [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2), [m3] "s"(m3),
                       [n0] "v"(nf.x), [n1] "v"(nf.y), [n2] "v"(nf.z), [n3] "v"(nf.w), [b0] "v"(b.x), [b1] "v"(b.y), [b2] "v"(b.z), [b3] "v"(b.w));
    }

    // Per-thread LDS float offsets, fixed for the whole sweep (slot and tile fold into the instructions' offset fields).
    struct Addr {
        int b;         // B operands of tile x, all four k-chunks: ct[.][16 x + p][4 g ..]   (+ 16 CS x)
        int bown;      // the same for this wave's own columns
        int piece;     // ct[.][16 w + 4 g + y][p]   (+ CS y)
        __device__ __forceinline__ void init() {
            const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4, w = wave();
            b = p * CS + 4 * g;
            bown = (16 * w + p) * CS + 4 * g;
            piece = (16 * w + 4 * g) * CS + p;
        }
    };

    __device__ static __forceinline__ f32x4_t mfma(float a, float b, f32x4_t c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    __device__ static __forceinline__ float4 ld4(const float* q) { return *reinterpret_cast<const float4*>(q); }

    // This wave's 16 columns of the pivot rows of block step X, out of tile X (see the header), into slot SW; the owner's carry
    // E - I at the pivot columns.
    template <int X, int SW>
    __device__ static __forceinline__ void piece(const f32x4_t (&acc)[8], const Addr& ad, Smem& sm) {
        const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4;
        f32x4_t v = acc[X];
        if (wave() == X && (p >> 2) == g) {
            const int a = p & 3;
            v.x -= (a == 0) ? 1.f : 0.f; v.y -= (a == 1) ? 1.f : 0.f; v.z -= (a == 2) ? 1.f : 0.f; v.w -= (a == 3) ? 1.f : 0.f;
        }
        float* q = &sm.ct[SW][0][0] + ad.piece;
        q[0] = v.x; q[CS] = v.y; q[2 * CS] = v.z; q[3 * CS] = v.w;
    }

    // Sub-step SB of the owner's in-tile sweep: the 4 x 4 block D of the tile's rows 4 SB .. 4 SB + 3 (lane group SB, one quad)
    // is inverted by gj4; the rank-4 update of the whole tile is one MFMA whose operands come off a 320-byte turn-table in LDS
    // that only this wave touches (a wave's LDS instructions execute in order: no barrier, no wait beyond the reads' own).
    template <int SB>
    __device__ static __forceinline__ void sub_prepare(f32x4_t& t, float (&D)[4], float4& c4, Smem& sm) {
        const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4, a = p & 3;
        const bool rowg = g == SB, dq = rowg && (p >> 2) == SB;
        D[0] = t.x; D[1] = t.y; D[2] = t.z; D[3] = t.w;          // quad lane a: column a of D = row a
        f32x4_t c = t;                                             // C': D - I at the pivot columns
        c.x -= (dq && a == 0) ? 1.f : 0.f; c.y -= (dq && a == 1) ? 1.f : 0.f; c.z -= (dq && a == 2) ? 1.f : 0.f; c.w -= (dq && a == 3) ? 1.f : 0.f;
        if (rowg) *reinterpret_cast<float4*>(&sm.cw[p][0]) = make_float4(c.x, c.y, c.z, c.w);
#if ADKF_M16_TURN
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        c4 = ld4(&sm.cw[p][0]);
#endif
        // M_PP := D - 2I
        t.x -= (dq && a == 0) ? 2.f : 0.f; t.y -= (dq && a == 1) ? 2.f : 0.f; t.z -= (dq && a == 2) ? 2.f : 0.f; t.w -= (dq && a == 3) ? 2.f : 0.f;
    }
    // D^-1 and the pivots out (the quad's four lanes), the turn-table reads in flight
    template <int SB, int X>
    __device__ static __forceinline__ void sub_turn(const float (&D)[4], const float (&piv)[4], float4& c4, float4& d4, Smem& sm) {
        const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4, a = p & 3;
        const bool dq = g == SB && (p >> 2) == SB;
#if ADKF_M16_TURN
        if (dq && a == 0) *reinterpret_cast<float4*>(&sm.pivs[16 * X + 4 * SB]) = make_float4(piv[0], piv[1], piv[2], piv[3]);
        // G[j][k] = D[k] of lane 20 SB + j; this lane needs row g: r_j = sum_k G[j][k] c4[k], selected by g
        float r[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float g0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, D[0]), 20 * SB + j));
            const float g1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, D[1]), 20 * SB + j));
            const float g2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, D[2]), 20 * SB + j));
            const float g3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, D[3]), 20 * SB + j));
            r[j] = fmaf(g0, c4.x, fmaf(g1, c4.y, fmaf(g2, c4.z, g3 * c4.w)));
        }
        d4 = make_float4(r[0], r[1], r[2], r[3]);
#else
        if (dq) {
            *reinterpret_cast<float4*>(&sm.dw[a][0]) = make_float4(D[0], D[1], D[2], D[3]);
            if (a == 0) *reinterpret_cast<float4*>(&sm.pivs[16 * X + 4 * SB]) = make_float4(piv[0], piv[1], piv[2], piv[3]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        c4 = ld4(&sm.cw[p][0]);
        d4 = ld4(&sm.dw[g][0]);
#endif
    }
    __device__ static __forceinline__ void sub_update(f32x4_t& t, const float4& c4, const float4& d4) {
        const int g = (threadIdx.x & 63) >> 4;
#if ADKF_M16_TURN
        const float av = -(g == 0 ? d4.x : g == 1 ? d4.y : g == 2 ? d4.z : d4.w);
#else
        const float av = -fmaf(d4.x, c4.x, fmaf(d4.y, c4.y, fmaf(d4.z, c4.z, d4.w * c4.w)));
#endif
        const float bv = g == 0 ? c4.x : g == 1 ? c4.y : g == 2 ? c4.z : c4.w;
        t = mfma(av, bv, t);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");   // the next sub-step's stores stay behind these reads
        __builtin_amdgcn_wave_barrier();
    }
    // one whole sub-step without any passenger (the first block step's owner, before the loop)
    template <int SB, int X>
    __device__ static __forceinline__ void sub_plain(f32x4_t& t, Smem& sm) {
        float D[4], piv[4];
        float4 c4, d4;
        sub_prepare<SB>(t, D, c4, sm);
        if (!(ADKF_M_ABLATE & 2)) gj4(D, piv); else { piv[0] = piv[1] = piv[2] = piv[3] = 1.f; }
        sub_turn<SB, X>(D, piv, c4, d4, sm);
        sub_update(t, c4, d4);
    }

    // -E^-1 (the swept tile X of wave X) into slot SW: einv[m = p][k = 4 g + y] (symmetric)
    template <int X, int SW>
    __device__ static __forceinline__ void publish_einv(const f32x4_t (&acc)[8], Smem& sm) {
        const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4;
        *reinterpret_cast<float4*>(&sm.einv[SW][p][4 * g]) = make_float4(acc[X].x, acc[X].y, acc[X].z, acc[X].w);
    }

    template <int S, int J>
    __device__ static __forceinline__ bool takes(int w) { return !(J == S && w == S); }   // the owner's diagonal tile holds -E^-1 already

    // k-chunk Y of the update of tile J (the components of a float4 / f32x4_t by a compile-time index)
    template <int Y> __device__ static __forceinline__ float comp(const float4& v) { return Y == 0 ? v.x : Y == 1 ? v.y : Y == 2 ? v.z : v.w; }
    template <int Y> __device__ static __forceinline__ float comp(const f32x4_t& v) { return Y == 0 ? v.x : Y == 1 ? v.y : Y == 2 ? v.z : v.w; }
    template <int J, int Y>
    __device__ static __forceinline__ void upd(f32x4_t (&acc)[8], const f32x4_t& nf, const float4 (&b4)[8]) {
#if !(ADKF_M_ABLATE & 8)
        acc[J] = mfma(comp<Y>(nf), comp<Y>(b4[J]), acc[J]);
#endif
    }
    // entries K0 .. K1 - 1 of the chain wave's passenger list in the LDS waits of its sub-steps: tiles T2, T4, T6 (T_i = NX + i),
    // four chunks each
    template <int NX, int K0, int K1>
    __device__ static __forceinline__ void riders(f32x4_t (&acc)[8], const f32x4_t& nf, const float4 (&b4)[8]) {
        if constexpr (K0 < K1) {
            upd<(NX + 2 + 2 * (K0 >> 2)) & 7, K0 & 3>(acc, nf, b4);
            riders<NX, K0 + 1, K1>(acc, nf, b4);
        }
    }

    // Sub-step SB of the chain wave inside block step S: tile T_(2 SB + 1)'s update rides in the Gauss-Jordan stream, three more
    // MFMAs go out while the turn-table reads are in flight.
    template <int S, int SB>
    __device__ static __forceinline__ void sub_chain(f32x4_t (&acc)[8], const f32x4_t& nf, float4 (&b4)[8], const Addr& ad, Smem& sm) {
        constexpr int NX = (S + 1) & 7, SLOT = S & 1, TG = (NX + 1 + 2 * SB) & 7;
        float D[4], piv[4];
        float4 c4, d4;
        sub_prepare<SB>(acc[NX], D, c4, sm);
        __builtin_amdgcn_sched_barrier(0);
#if (ADKF_M_ABLATE & 2)
        piv[0] = piv[1] = piv[2] = piv[3] = 1.f;
        upd<TG, 0>(acc, nf, b4); upd<TG, 1>(acc, nf, b4); upd<TG, 2>(acc, nf, b4); upd<TG, 3>(acc, nf, b4);
#elif (ADKF_M_ABLATE & (8 | 32))
        gj4(D, piv);
#else
        gj4m(D, piv, acc[TG], nf, b4[TG]);
#endif
        __builtin_amdgcn_sched_barrier(0);
        sub_turn<SB, NX>(D, piv, c4, d4, sm);
        if (SB == 0 && !(ADKF_M_ABLATE & 32)) {   // the B operands of the tiles that have not been fetched yet queue up behind the turn-table reads
            const float* ctr = &sm.ct[SLOT][0][0];
#pragma unroll
            for (int i = 3; i <= 7; ++i) b4[(NX + i) & 7] = ld4(ctr + ad.b + 16 * CS * ((NX + i) & 7));
        }
        __builtin_amdgcn_sched_barrier(0);
        sub_update(acc[NX], c4, d4);
        __builtin_amdgcn_sched_barrier(0);
        // three more MFMAs BEHIND the tile's own: the matrix pipe takes a SIMD's MFMAs in the order they were issued, so
        // whatever goes in first is what the chain waits for
        if (!(ADKF_M_ABLATE & 32)) riders<NX, 3 * SB, 3 * SB + 3>(acc, nf, b4);
        __builtin_amdgcn_sched_barrier(0);
    }

    // Block step S.  On entry (behind the barrier) slot S & 1 holds C' and -E^-1 of the step; on exit slot (S + 1) & 1 holds those of
    // step S + 1 and every tile has the update of step S.
    template <int S>
    __device__ static __forceinline__ void step(f32x4_t (&acc)[8], const Addr& ad, Smem& sm) {
        constexpr int SLOT = S & 1, NX = (S + 1) & 7;
        constexpr bool LAST = S == 7;
        const int w = wave();
        const bool is_chain = !LAST && w == NX;
        ADKF_MTS(0);
        if (is_chain) __builtin_amdgcn_s_setprio(3);
        const float* ctr = &sm.ct[SLOT][0][0];
        const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4;
        // the operands of the hand-off chain first: -E^-1, the wave's own columns, the critical tile's B (+ two tiles' worth for
        // the chain wave's first passengers); everything else after the piece is out
        const float4 a4 = ld4(&sm.einv[SLOT][p][4 * g]);
        const float4 c4 = ld4(ctr + ad.bown);
        float4 b4[8];
        if (!LAST) {
            b4[NX] = ld4(ctr + ad.b + 16 * CS * NX);
            b4[(NX + 1) & 7] = ld4(ctr + ad.b + 16 * CS * ((NX + 1) & 7));
            b4[(NX + 2) & 7] = ld4(ctr + ad.b + 16 * CS * ((NX + 2) & 7));
        }
        __builtin_amdgcn_sched_barrier(0);
        // -F for this wave's columns: two chains of two MFMAs
        f32x4_t nf = {0.f, 0.f, 0.f, 0.f}, nf1 = {0.f, 0.f, 0.f, 0.f};
        nf = mfma(a4.x, c4.x, nf); nf1 = mfma(a4.z, c4.z, nf1); nf = mfma(a4.y, c4.y, nf); nf1 = mfma(a4.w, c4.w, nf1);
        nf += nf1;
        ADKF_MTS(1);
        if (LAST) {
#pragma unroll
            for (int x = 0; x < 8; ++x) b4[x] = ld4(ctr + ad.b + 16 * CS * x);
#define ADKF_M16_ALL(Y) do { upd<0, Y>(acc, nf, b4); upd<1, Y>(acc, nf, b4); upd<2, Y>(acc, nf, b4); upd<3, Y>(acc, nf, b4); \
                             upd<4, Y>(acc, nf, b4); upd<5, Y>(acc, nf, b4); upd<6, Y>(acc, nf, b4); if (w != 7) upd<7, Y>(acc, nf, b4); } while (0)
            ADKF_M16_ALL(0); ADKF_M16_ALL(1); ADKF_M16_ALL(2); ADKF_M16_ALL(3);
            return;
        }
        {   // the tile the next hand-off comes out of: again two chains of two
            f32x4_t t1 = {0.f, 0.f, 0.f, 0.f};
            acc[NX] = mfma(nf.x, b4[NX].x, acc[NX]); t1 = mfma(nf.z, b4[NX].z, t1);
            acc[NX] = mfma(nf.y, b4[NX].y, acc[NX]); t1 = mfma(nf.w, b4[NX].w, t1);
            acc[NX] += t1;
        }
        __builtin_amdgcn_sched_barrier(0);
        piece<NX, SLOT ^ 1>(acc, ad, sm);
        ADKF_MTS(2);
        __builtin_amdgcn_sched_barrier(0);
        if (is_chain) {
            // the next E: swept in place, four sub-steps, this wave's own 28 bulk MFMAs riding along (sub_chain)
            sub_chain<S, 0>(acc, nf, b4, ad, sm); ADKF_MTS(3);
            sub_chain<S, 1>(acc, nf, b4, ad, sm); ADKF_MTS(7);
            sub_chain<S, 2>(acc, nf, b4, ad, sm); ADKF_MTS(8);
            sub_chain<S, 3>(acc, nf, b4, ad, sm);
            ADKF_MTS(4);
            publish_einv<NX, SLOT ^ 1>(acc, sm);
            ADKF_MTS(5);
            __builtin_amdgcn_s_setprio(0);
        } else {
#pragma unroll
            for (int i = 3; i <= 7; ++i) b4[(NX + i) & 7] = ld4(ctr + ad.b + 16 * CS * ((NX + i) & 7));
            __builtin_amdgcn_sched_barrier(0);
            // seven tiles, chunk-major (consecutive MFMAs independent); tile S of wave S holds -E^-1 already
            // (the chain wave's SIMD partner - waves w and w + 4 share a SIMD - spaces its MFMAs out: a SIMD's matrix pipe takes
            // MFMAs in issue order, and 28 of them issued back to back are 900 cycles the chain wave's own would queue behind)
            const bool paced = ADKF_M16_PACE && w == ((NX + 4) & 7);
#define ADKF_M16_GAP() do { __builtin_amdgcn_sched_barrier(0); if (paced) __builtin_amdgcn_s_sleep(ADKF_M16_PACE); __builtin_amdgcn_sched_barrier(0); } while (0)
#define ADKF_M16_BULK(Y) do { upd<(NX + 1) & 7, Y>(acc, nf, b4); upd<(NX + 2) & 7, Y>(acc, nf, b4); ADKF_M16_GAP(); upd<(NX + 3) & 7, Y>(acc, nf, b4); upd<(NX + 4) & 7, Y>(acc, nf, b4); ADKF_M16_GAP(); \
                              upd<(NX + 5) & 7, Y>(acc, nf, b4); upd<(NX + 6) & 7, Y>(acc, nf, b4); ADKF_M16_GAP(); if (w != S) upd<S, Y>(acc, nf, b4); } while (0)
            ADKF_M16_BULK(0); ADKF_M16_BULK(1); ADKF_M16_BULK(2); ADKF_M16_BULK(3);
        }
        ADKF_MTS(6);
    }

    // In: m = this thread's block of the SPD matrix (identity-padded beyond n).  Out: m = -(A^-1) on the leading n x n part;
    // the pivots are left in sm.pivs[0..128) (by matrix index; 1 for identity padding).  All threads call.
    __device__ static __forceinline__ void run(float (&m)[RB][CB], int n, Smem& sm) {
        const int wv = wave();
        f32x4_t acc[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) { acc[x].x = m[x][0]; acc[x].y = m[x][1]; acc[x].z = m[x][2]; acc[x].w = m[x][3]; }
        Addr ad;
        ad.init();
        if (n > 0) {
            // block step 0 into slot 0: every wave delivers its piece out of tile 0, wave 0 sweeps its diagonal tile
            piece<0, 0>(acc, ad, sm);
            if (wv == 0) {
                sub_plain<0, 0>(acc[0], sm); sub_plain<1, 0>(acc[0], sm); sub_plain<2, 0>(acc[0], sm); sub_plain<3, 0>(acc[0], sm);
                publish_einv<0, 0>(acc, sm);
            }
            __syncthreads(); step<0>(acc, ad, sm);
            __syncthreads(); step<1>(acc, ad, sm);
            __syncthreads(); step<2>(acc, ad, sm);
            __syncthreads(); step<3>(acc, ad, sm);
            __syncthreads(); step<4>(acc, ad, sm);
            __syncthreads(); step<5>(acc, ad, sm);
            __syncthreads(); step<6>(acc, ad, sm);
            __syncthreads(); step<7>(acc, ad, sm);
        }
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 8; ++x) { m[x][0] = acc[x].x; m[x][1] = acc[x].y; m[x][2] = acc[x].z; m[x][3] = acc[x].w; }
        // (opaque re-definition: see factor.h - keeps the consumers' pairing choices out of the sweep's register assignment)
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int c = 0; c < CB; ++c) asm volatile("" : "+v"(m[r][c]));
    }

    __device__ static __forceinline__ int finish(int n, Smem& sm, float& logdet) {
        const int tid = threadIdx.x;
        float v[1] = {0.f};
        int bad = INT_MAX;
        for (int k = tid; k < n; k += NT) {
            const float p = sm.pivs[k];
            v[0] += logf(p);
            if (!(p > 0.f) && k + 1 < bad) bad = k + 1;
        }
        block_sum<1, NT>(v, sm.red);
        logdet = v[0];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { const int other = __shfl_xor(bad, o, 64); bad = other < bad ? other : bad; }
        __syncthreads();
        if ((tid & 63) == 0) sm.redi[tid >> 6] = bad;
        __syncthreads();
        int info = INT_MAX;
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) info = sm.redi[w] < info ? sm.redi[w] : info;
        return info == INT_MAX ? 0 : info;
    }

    // out[j] = sum_i (-m_ij) in[i], i.e. A^-1 * in (the matrix is symmetric: the thread sums its 8 rows for each of its 4
    // columns, the 16 lanes that share the columns are one DPP row).  `in` must be visible (barrier before); `out` is visible
    // on return.
    __device__ static __forceinline__ void solve(const float (&m)[RB][CB], const float* in, float* out) {
        float x[RB], sc[CB];
#pragma unroll
        for (int r = 0; r < RB; ++r) x[r] = in[row(r)];
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            float acc = 0.f;
#pragma unroll
            for (int r = 0; r < RB; ++r) acc = fmaf(-m[r][c], x[r], acc);
            sc[c] = acc;
        }
#pragma unroll
        for (int c = 0; c < CB; ++c) {
            sc[c] += dpp_f<DPP_XOR1>(sc[c]);
            sc[c] += dpp_f<DPP_XOR2>(sc[c]);
            sc[c] += dpp_f<DPP_HALF_MIRROR>(sc[c]);
            sc[c] += dpp_f<DPP_MIRROR>(sc[c]);
        }
        if ((threadIdx.x & 15) == 0) *reinterpret_cast<float4*>(out + col(0)) = make_float4(sc[0], sc[1], sc[2], sc[3]);
        __syncthreads();
    }
};

}  // namespace adkf
