// Sweep<128, 512> on the matrix pipe: the symmetric sweep of factor.h with the rank-4 update of every block step issued as
// v_mfma_f32_16x16x4_f32 (exact fp32: bit-for-bit a k-ordered fmaf chain), the chain of a block step cut down to the 4 x 4
// inverse alone, and the updates of a step running one step BEHIND its hand-off (included at the end of factor.h; factor_w.h
// keeps the VALU variant for A/B runs, ADKF_SWEEP_M=0).
//
// Layout.  Wave w (0..7), lane l: p = l & 15, g = l >> 4.  The wave owns the 16 matrix rows 16 w .. 16 w + 15 as eight
// 16 x 16 accumulator tiles; tile x covers the columns 16 x .. 16 x + 15.  In the MFMA's C/D map lane l, register y of
// tile x is
//        internal row  I = 16 w + 4 g + y,        internal column  J = 16 x + p.
// The kernels around the sweep see the TRANSPOSED element (the matrix is symmetric on input and output): m[r = x][c = y]
// is row(r) = J, col(c) = I, i.e. a thread's block is RB = 8 rows (16 apart) x CB = 4 consecutive columns 4 (4 w + g) .. + 3.
//
// Block step s = 8 gq + wq sweeps the 4 x 4 diagonal block D of the pivot rows P = 16 wq + 4 gq + {0..3} (lane groups
// outermost: consecutive steps are owned by different waves; all 32 blocks are swept whatever n is - a block of identity
// padding is a no-op with pivots 1).  What a step hands over through LDS:
//   * the pivot rows C (with D - I at the pivot columns, factor.h's trick) as the [column][k] image `ct`.  NOBODY brings
//     128-wide rows up to date for that: every wave delivers its own 16 columns from the TRANSPOSED tile it holds (tile wq:
//     its rows x the pivot columns; lanes p = 4 gq + k of every lane group hold M[16 w + 4 g + y][P_k] = C[k][16 w + 4 g + y]
//     in registers y = 0..3), four dwords per lane of one quad per lane group;
//   * D^-1, from the owner: D sits in ONE QUAD of wave wq (lane group gq, lanes p = 4 gq + a, registers 0..3 of tile wq:
//     quad lane a holds row a) and is inverted in place by four Gauss-Jordan steps whose pivot row travels by DPP quad_perm
//     fused into v_fmac_f32_dpp - no ds_bpermute, no readlane, and no elimination of the 128-wide rows: F = D^-1 C is formed
//     by whoever needs it, ONE value per lane (the MFMA's A operand A[m = p][k = g] = -F[g][16 w + p]).
// A consumer's B operand of tile x - B[k = g][n = p] = C[g][16 x + p] - is one conflict-free ds_read_b32.
//
// Schedule.  tools/handoff_bench.hip and the ablation builds of tools/sweepm_bench.hip priced a block step in which every
// wave fetches, updates its eight tiles and then publishes: barrier + LDS round trip + operands ~500 cycles, eight MFMAs per
// wave ~590 (the SIMD's matrix pipe: 16 x 32), piece + LDS write latency ~170, the inversion ~320 - in SERIES, ~1600 per step.
// Only ONE tile per wave is on the critical path of the next hand-off (tile wq' of the next block: it yields the wave's piece
// and, in the owner, D).  So a step applies its own update to that tile alone and the update of the PREVIOUS step to the other
// seven, from operands that are already in registers: those seven MFMAs issue right behind the barrier, under the LDS reads
// of the new vectors, and the hand-off chain (reads -> A operand -> one MFMA -> piece / inversion -> LDS write -> barrier)
// no longer waits for the matrix pipe.  Everything is unrolled over the eight steps of a lane group, so tile indices, slots
// and LDS offsets are compile-time constants; the first step's "previous" operands are zeros, and the last step's update is
// flushed after the loop.  Two vector slots: the writers of step s fill slot (s + 1) & 1 while everybody reads slot s & 1,
// whose previous readers (step s - 1) passed the barrier of step s with their reads complete.
//
// Who goes first behind the barrier (round 3, second session).  A SIMD's matrix pipe takes the MFMAs of its two waves in the
// order they were issued, and the LDS serves reads in the order they arrive: whatever the other waves put in front of the
// chain wave's three critical reads and its one critical MFMA is what the hand-off waits for.  Only the next owner is on the
// chain - the others' pieces are not needed before the next barrier, i.e. they have the inversion's ~190 cycles of slack - so
// every wave but the chain wave sleeps 128 cycles behind the barrier (ADKF_M_SLEEP): 37.2 k -> 35.0 k cycles per sweep; with
// only the chain wave's SIMD partner (wave + 4) sleeping 35.6 k, i.e. most of it is the partner's early MFMAs.
#pragma once
#ifndef ADKF_M_ABLATE
#define ADKF_M_ABLATE 0   // timing-only ablations for tools/sweepm_bench.hip (2: no inversion, 8: no MFMA at all, 16: one B read instead of eight, 32: no owner work at all, 64: no pieces)
#endif
#ifndef ADKF_M_EARLY
#define ADKF_M_EARLY 2    // how many of the previous step's seven MFMAs issue before the A operand of the new step is formed
#endif
#ifndef ADKF_M_A128
#define ADKF_M_A128 1     // 1: the A operand from two 16-byte LDS reads and four FMAs (36.2 k cycles per sweep); 0: from two dword reads, the row swaps of gfx950 and four DPP FMAs (37.9 k: the swap sequence with its hazard waits is on the hand-off chain)
#endif
#ifndef ADKF_M_SLEEP
#define ADKF_M_SLEEP 2    // s_sleep argument (x 64 cycles) of the waves that are NOT on the hand-off chain, right behind the barrier: 37.2 k -> 35.0 k cycles per sweep (1: 37.0 k, 3: 35.8 k)
#endif
#ifndef ADKF_M_SLEEP_MODE
#define ADKF_M_SLEEP_MODE 0   // who sleeps.  0: every wave but the chain wave (35.0 k); 1: only the chain wave's SIMD partner, wave + 4 (35.6 k: most of the gain is there); 2: everybody but the chain wave, behind its LDS reads instead of in front of them (37.0 k: no gain)
#endif
#ifndef ADKF_M_MID
#define ADKF_M_MID 1      // ... and between the critical MFMA and the piece that is read out of its result (they cover its latency)
#endif

namespace adkf {

typedef float f32x4_t __attribute__((ext_vector_type(4)));

template <> struct SweepSmem<128, 512> {
    static constexpr int NSLOT = 2;
    alignas(16) float ct[NSLOT][128][4];     // C^T: [column][k], D - I at the pivot columns
    alignas(16) float dinv[NSLOT][4][4];     // D^-1
    alignas(16) float pivs[128];
    alignas(16) float vec_in[128];
    alignas(16) float vec_out[128];
    float red[8 * 8];
    int redi[8];
#if ADKF_STAMP
    unsigned long long stamp[8 * 16];
#endif
    static constexpr int SCRATCH_FLOATS = NSLOT * 128 * 4;
    __device__ __forceinline__ float* scratch() { return &ct[0][0][0]; }   // free for the caller between two sweeps
};

#if ADKF_STAMP   // diagnostic build (tools/sweepm_bench.hip -DADKF_STAMP=<step> -DADKF_STAMP_SITE=<k>): s_memtime after the barrier of block
                 // step <step> and at ONE further site k per build - a stamp has to wait for its own result (lgkmcnt), which drains
                 // the wave's LDS queue and moves everything behind it, so two sites in one build would not be independent
#ifndef ADKF_STAMP_SITE
#define ADKF_STAMP_SITE 1
#endif
#define ADKF_MTS(slot_) do { if ((slot_ == 0 || slot_ == ADKF_STAMP_SITE) && s_stamp == ADKF_STAMP && (threadIdx.x & 63) == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); sm.stamp[(threadIdx.x >> 6) * 16 + (slot_)] = t_; } } while (0)
#else
#define ADKF_MTS(slot_) do {} while (0)
#endif

template <> struct Sweep<128, 512> {
    using Smem = SweepSmem<128, 512>;
    static constexpr int NMAX = 128, NT = 512, RB = 8, CB = 4, B = 4, NW = 8;

    __device__ static __forceinline__ int wave() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }
    __device__ static __forceinline__ int bc() { return threadIdx.x >> 4; }   // column block: 4 w + g
    __device__ static __forceinline__ int row(int r) { return (r << 4) + (threadIdx.x & 15); }
    __device__ static __forceinline__ int col(int c) { return bc() * CB + c; }

    template <int P> __device__ static __forceinline__ float quad_bcast(float v) {
        return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), P * 0x55, 0xF, 0xF, true));
    }

    // In-place inverse of the 4 x 4 block D inside every quad (quad lane a holds row a in d[0..3]) by four Gauss-Jordan steps,
    // and its successive pivots (uniform in the quad).  Step P: row P is scaled by r = 1 / d_PP, row a loses d_aP r times row P,
    // column P becomes the multipliers; "x += g * (x of quad lane P)" runs as v_fmac_f32 with the DPP quad broadcast on its first
    // source (hipcc does not fold a mov_dpp into the FMA), with the uniform coefficient g = r - 1 on the pivot lane and -d_aP r on
    // the others formed as ONE fma(-r, y, z) whose y and z are selected while the reciprocal is in flight.  This is the serial
    // core of every block step (all 128 pivots pass through it), so it is one hand-scheduled instruction stream: the dependent
    // chain of a pivot is mov_dpp -> rcp -> (Newton: 2 fma) -> fma -> fmac_dpp, the next pivot's column is updated first, and
    // the "VALU write -> DPP read of the same VGPR" wait states (2) are always covered by the two instructions in between -
    // no s_nop but the leading one (which covers whatever the compiler placed last).
#ifndef ADKF_M_NEWTON
#define ADKF_M_NEWTON 0   // 1: one Newton step on the pivot reciprocals (v_rcp_f32 is 1 ulp; the inverses and log-determinants of tools/sweepm_bench.hip are the same to three digits either way, the step costs 50 cycles per block step)
#endif
#if !ADKF_M_NEWTON
#define ADKF_MGJ_NEWTON(pv)
#else
#define ADKF_MGJ_NEWTON(pv) "v_fma_f32 %[e], -" pv ", %[r], 1.0\n\tv_fma_f32 %[r], %[e], %[r], %[r]\n\t"
#endif
#define ADKF_MGJ_STEP(P, dP, dA, dB, dC, pv, mk) \
        "v_mov_b32_dpp " pv ", " dP " quad_perm:[" #P "," #P "," #P "," #P "] row_mask:0xf bank_mask:0xf\n\t" \
        "v_rcp_f32_e32 %[r], " pv "\n\t" \
        "v_cndmask_b32_e64 %[y], " dP ", -1.0, " mk "\n\t" \
        "v_cndmask_b32_e64 %[z], 0, -1.0, " mk "\n\t" \
        ADKF_MGJ_NEWTON(pv) \
        "v_fma_f32 %[g], -%[r], %[y], %[z]\n\t" \
        "v_fmac_f32_dpp " dA ", " dA ", %[g] quad_perm:[" #P "," #P "," #P "," #P "] row_mask:0xf bank_mask:0xf\n\t" \
        "v_fmac_f32_dpp " dB ", " dB ", %[g] quad_perm:[" #P "," #P "," #P "," #P "] row_mask:0xf bank_mask:0xf\n\t" \
        "v_fmac_f32_dpp " dC ", " dC ", %[g] quad_perm:[" #P "," #P "," #P "," #P "] row_mask:0xf bank_mask:0xf\n\t" \
        "v_cndmask_b32_e64 " dP ", %[g], %[r], " mk "\n\t"
    __device__ static __forceinline__ void gj4(float (&D)[4], float (&piv)[4]) {
        float r, e, y, z, g;
        const unsigned long long m0 = 0x1111111111111111ull, m1 = 0x2222222222222222ull, m2 = 0x4444444444444444ull, m3 = 0x8888888888888888ull;
        asm volatile("s_nop 1\n\t"
                     ADKF_MGJ_STEP(0, "%[d0]", "%[d1]", "%[d2]", "%[d3]", "%[p0]", "%[m0]")
                     ADKF_MGJ_STEP(1, "%[d1]", "%[d2]", "%[d3]", "%[d0]", "%[p1]", "%[m1]")
                     ADKF_MGJ_STEP(2, "%[d2]", "%[d3]", "%[d0]", "%[d1]", "%[p2]", "%[m2]")
                     ADKF_MGJ_STEP(3, "%[d3]", "%[d0]", "%[d1]", "%[d2]", "%[p3]", "%[m3]")
                     : [d0] "+v"(D[0]), [d1] "+v"(D[1]), [d2] "+v"(D[2]), [d3] "+v"(D[3]),
                       [p0] "=&v"(piv[0]), [p1] "=&v"(piv[1]), [p2] "=&v"(piv[2]), [p3] "=&v"(piv[3]),
                       [r] "=&v"(r), [e] "=&v"(e), [y] "=&v"(y), [z] "=&v"(z), [g] "=&v"(g)
                     : [m0] "s"(m0), [m1] "s"(m1), [m2] "s"(m2), [m3] "s"(m3));
        (void)e;
    }

    struct Ops { float a; float b[8]; };   // the operands of one block step's update: A and the eight B

    // Per-thread LDS float offsets, fixed for the whole sweep (slot and tile are compile-time constants at every use and fold
    // into the instructions' offset fields: a block step computes no address).
    struct Addr {
        int b;         // B operand of tile x: ct[.][16 x + p][g]  (+ 64 x)
        int bd;        // the same for this wave's own columns 16 w + p: carries the pivot-row values of the A operand
        int dq;        // dinv[.][g][p & 3]
        int piece;     // ct[.][16 w + 4 g + y][p & 3]  (+ 4 y)
        __device__ __forceinline__ void init() {
            const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4, w = wave();
            b = p * 4 + g;
            bd = (16 * w + p) * 4 + g;
            dq = g * 4 + (p & 3);
            piece = (16 * w + 4 * g) * 4 + (p & 3);
        }
    };

    template <int X>
    __device__ static __forceinline__ void mfma(f32x4_t (&acc)[8], float a, float b) {
#if !(ADKF_M_ABLATE & 8)
        acc[X] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[X], 0, 0, 0);
#endif
    }
    // the I-th (0..6) tile of the previous step's update in a step whose critical tile is WN and whose predecessor's was K = WN - 1:
    // every tile but K, starting with WN (the critical MFMA accumulates on top of it)
    template <int WN, int I> static constexpr int bulk_tile() { return (WN + I + (I >= 7 ? 1 : 0)) & 7; }   // WN, WN+1, .., WN+6 (WN+7 = K is skipped)
    template <int WN, int I0, int I1>
    __device__ static __forceinline__ void bulk(f32x4_t (&acc)[8], const Ops& prev) {
        if constexpr (I0 < I1) {
            mfma<bulk_tile<WN, I0>()>(acc, prev.a, prev.b[bulk_tile<WN, I0>()]);
            bulk<WN, I0 + 1, I1>(acc, prev);
        }
    }

    // A[m = p][k = g] = -F[g][16 w + p] = -sum_k' dinv[g][k'] C[k'][16 w + p] from one dword of D^-1 per lane (lane (g, p)
    // holds dinv[g][p & 3]: the row is rebuilt by quad broadcasts) and the B operand of the wave's own columns, which carries
    // C[g][16 w + p]: the other three lane groups' values come by the row swaps of gfx950 (v_permlane16/32_swap) instead of a
    // 16-byte LDS read.  (In assembly: given the builtins, hipcc took the two results of a swap for equal and dropped one; the
    // s_nop cover the VALU write -> permlane swap read wait states the compiler cannot see in here, the last one the VALU
    // write -> MFMA operand read.)
    __device__ static __forceinline__ float a_operand(float dq, float bd) {
        float c0, c1, c2, c3, av;
        asm volatile("v_mov_b32 %[c0], %[b0]\n\tv_mov_b32 %[c1], %[b0]\n\ts_nop 1\n\t"
                     "v_permlane16_swap_b32 %[c0], %[c1]\n\ts_nop 1\n\t"          // c0 = [r0 r0 r2 r2], c1 = [r1 r1 r3 r3]
                     "v_mov_b32 %[c2], %[c0]\n\tv_mov_b32 %[c3], %[c1]\n\ts_nop 1\n\t"
                     "v_permlane32_swap_b32 %[c0], %[c2]\n\t"                      // C[0][.] everywhere, C[2][.] everywhere
                     "v_permlane32_swap_b32 %[c1], %[c3]\n\ts_nop 1"
                     : [c0] "=&v"(c0), [c1] "=&v"(c1), [c2] "=&v"(c2), [c3] "=&v"(c3) : [b0] "v"(bd));
        asm volatile("v_mul_f32_dpp %[t], -%[d], %[c0] quad_perm:[0,0,0,0] row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %[t], -%[d], %[c1] quad_perm:[1,1,1,1] row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %[t], -%[d], %[c2] quad_perm:[2,2,2,2] row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f32_dpp %[t], -%[d], %[c3] quad_perm:[3,3,3,3] row_mask:0xf bank_mask:0xf\n\ts_nop 1"
                     : [t] "=&v"(av) : [d] "v"(dq), [c0] "v"(c0), [c1] "v"(c1), [c2] "v"(c2), [c3] "v"(c3));
        return av;
    }

    // This wave's 16 columns of the pivot rows of block (gn, WN), out of tile WN (see the header), into slot SW.
    template <int WN, int SW>
    __device__ static __forceinline__ void piece(const f32x4_t (&acc)[8], const Addr& ad, int gn, Smem& sm) {
        const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4, a = p & 3;
        if ((p >> 2) == gn && !(ADKF_M_ABLATE & 64)) {
            f32x4_t v = acc[WN];
            if (wave() == WN && g == gn) {   // C := D - I at the pivot columns (M_PP := D - 2I is additive and nobody reads M_PP again: run() applies it at the end)
                v.x -= (a == 0) ? 1.f : 0.f; v.y -= (a == 1) ? 1.f : 0.f; v.z -= (a == 2) ? 1.f : 0.f; v.w -= (a == 3) ? 1.f : 0.f;
            }
            float* q = &sm.ct[SW][0][0] + ad.piece;
            q[0] = v.x; q[4] = v.y; q[8] = v.z; q[12] = v.w;
        }
    }
    // the owner's D^-1 and pivots into slot SW (all lanes of wave WN call)
    template <int WN, int SW>
    __device__ static __forceinline__ void publish_dinv(const float (&D)[4], const float (&piv)[4], int gn, Smem& sm) {
        const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4, a = p & 3;
        if (g == gn && (p >> 2) == gn) {
            *reinterpret_cast<float4*>(&sm.dinv[SW][a][0]) = make_float4(D[0], D[1], D[2], D[3]);
            if (a == 0) {
                // (the index is formed HERE from an opaque copy of gn: hoisted out of the unrolled sweep this uniform address
                // sat in a VGPR that the 128-register build of k_inner spilled - one scratch reload per block step, on the
                // owner's way to the barrier)
                int gi = gn;
                asm volatile("" : "+s"(gi));
                *reinterpret_cast<float4*>(&sm.pivs[16 * WN + 4 * gi]) = make_float4(piv[0], piv[1], piv[2], piv[3]);
            }
        }
    }

    // Block step s = 8 gq + K (see the header).  `prev`: the operands of step s - 1 (zeros in the very first step); on return
    // the operands of this step, whose update has reached tile WN only.
    template <int K>
    __device__ static __forceinline__ void step(f32x4_t (&acc)[8], Ops& prev, const Addr& ad, int gq, Smem& sm, int s_stamp = -1) {
        (void)s_stamp;
        constexpr int SLOT = K & 1, WN = (K + 1) & 7;
        const int w = wave();
        const int gn = gq + (K == 7 ? 1 : 0);
        const bool has_next = gn < 4;
        const bool is_chain = has_next && w == WN && !(ADKF_M_ABLATE & 32);
        ADKF_MTS(0);
        if (is_chain) __builtin_amdgcn_s_setprio(3);
#if ADKF_M_SLEEP && ADKF_M_SLEEP_MODE == 0
        else __builtin_amdgcn_s_sleep(ADKF_M_SLEEP);
#elif ADKF_M_SLEEP && ADKF_M_SLEEP_MODE == 1
        else if (w == ((WN + 4) & 7)) __builtin_amdgcn_s_sleep(ADKF_M_SLEEP);
#endif
        // the operands of the hand-off chain first: D^-1, the wave's own columns, the critical tile's B; then the other seven
        const float* ctr = &sm.ct[SLOT][0][0];
#if ADKF_M_A128
        const float4 dv4 = *reinterpret_cast<const float4*>(&sm.dinv[SLOT][(threadIdx.x & 63) >> 4][0]);
        const float4 cr4 = *reinterpret_cast<const float4*>(ctr + (ad.bd & ~3));
#else
        const float dq = (&sm.dinv[SLOT][0][0])[ad.dq];
        const float bd = ctr[ad.bd];
#endif
        Ops cur;
        cur.b[WN] = ctr[ad.b + 64 * WN];
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int x = 0; x < 8; ++x)
            if (x != WN) cur.b[x] = (ADKF_M_ABLATE & 16) ? cur.b[WN] * (1.f + x) : ctr[ad.b + 64 * x];
        __builtin_amdgcn_sched_barrier(0);   // all ten reads in flight; the matrix pipe gets the previous step's update meanwhile
#if ADKF_M_SLEEP && ADKF_M_SLEEP_MODE == 2
        if (!is_chain) __builtin_amdgcn_s_sleep(ADKF_M_SLEEP);
        __builtin_amdgcn_sched_barrier(0);
#endif
        bulk<WN, 0, ADKF_M_EARLY>(acc, prev);
        __builtin_amdgcn_sched_barrier(0);
#if ADKF_M_A128
        cur.a = -fmaf(dv4.x, cr4.x, fmaf(dv4.y, cr4.y, fmaf(dv4.z, cr4.z, dv4.w * cr4.w)));
#else
        cur.a = a_operand(dq, bd);
#endif
        ADKF_MTS(1);
        mfma<WN>(acc, cur.a, cur.b[WN]);     // this step's update of the tile the next hand-off comes out of
        __builtin_amdgcn_sched_barrier(0);
        bulk<WN, ADKF_M_EARLY, ADKF_M_EARLY + ADKF_M_MID>(acc, prev);
        __builtin_amdgcn_sched_barrier(0);
        if (has_next) {
            piece<WN, SLOT ^ 1>(acc, ad, gn, sm);
            ADKF_MTS(2);
            if (is_chain) {
                float D[4] = {acc[WN].x, acc[WN].y, acc[WN].z, acc[WN].w}, piv[4] = {1.f, 1.f, 1.f, 1.f};
                if (!(ADKF_M_ABLATE & 2)) gj4(D, piv);
                ADKF_MTS(4);
                publish_dinv<WN, SLOT ^ 1>(D, piv, gn, sm);
                ADKF_MTS(5);
                __builtin_amdgcn_s_setprio(0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        bulk<WN, ADKF_M_EARLY + ADKF_M_MID, 7>(acc, prev);   // the rest issues under the latency of the LDS stores above
        // (no scheduling fence here: hipcc sinks two of them behind the barrier, in front of the next step's LDS reads, and the
        // sweep is 3 % FASTER that way than with all of them held in front of it - 37.3 k against 38.9 k cycles)
        prev = cur;
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
        Ops prev;
        prev.a = 0.f;
#pragma unroll
        for (int x = 0; x < 8; ++x) prev.b[x] = 0.f;
        if (n > 0) {
            // block (0, 0) into slot 0: every wave delivers its piece out of tile 0, wave 0 inverts
            piece<0, 0>(acc, ad, 0, sm);
            if (wv == 0) {
                float D[4] = {acc[0].x, acc[0].y, acc[0].z, acc[0].w}, piv[4] = {1.f, 1.f, 1.f, 1.f};
                gj4(D, piv);
                publish_dinv<0, 0>(D, piv, 0, sm);
            }
            for (int gq = 0; gq < 4; ++gq) {
                __syncthreads(); step<0>(acc, prev, ad, gq, sm, 8 * gq + 0);
                __syncthreads(); step<1>(acc, prev, ad, gq, sm, 8 * gq + 1);
                __syncthreads(); step<2>(acc, prev, ad, gq, sm, 8 * gq + 2);
                __syncthreads(); step<3>(acc, prev, ad, gq, sm, 8 * gq + 3);
                __syncthreads(); step<4>(acc, prev, ad, gq, sm, 8 * gq + 4);
                __syncthreads(); step<5>(acc, prev, ad, gq, sm, 8 * gq + 5);
                __syncthreads(); step<6>(acc, prev, ad, gq, sm, 8 * gq + 6);
                __syncthreads(); step<7>(acc, prev, ad, gq, sm, 8 * gq + 7);
            }
            bulk<1, 0, 7>(acc, prev);   // the last step's update: tile 0 has it (its "critical" tile), tiles 1..7 get it here
            // M_PP := D - 2I of every block, deferred: inside the loop the accumulators are written by nothing but MFMAs
            const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4;
            const bool dq = (p >> 2) == g;
#pragma unroll
            for (int x = 0; x < 8; ++x) {
                const bool mine = dq && wv == x;
                acc[x].x -= (mine && (p & 3) == 0) ? 2.f : 0.f; acc[x].y -= (mine && (p & 3) == 1) ? 2.f : 0.f;
                acc[x].z -= (mine && (p & 3) == 2) ? 2.f : 0.f; acc[x].w -= (mine && (p & 3) == 3) ? 2.f : 0.f;
            }
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
