// Sweep<128, 512>: the symmetric sweep of factor.h re-laid-out so that ONE WAVE owns each 4-pivot block entirely and the
// chain of a block step runs on quad-lane DPP instead of an explicit 4 x 4 inverse (included at the end of factor.h).
//
// Layout (wave w = tid >> 6, lane l: a = l & 3, cb = l >> 2): the thread holds the 4 x 8 block
//        rows  16 w + 4 a + r   (r = 0..3)          columns  8 cb + c   (c = 0..7)
// of the symmetric matrix.  Pivot block q (32 of them) is the four rows {16 w + 4 a + g : a = 0..3} with w = q mod 8,
// g = q / 8: register row g of ALL 64 lanes of wave w, one matrix row per quad lane.  The sweep order is therefore
// 0, 16, 32, .., 4, 20, ..: any order of pivots gives the same inverse, the pivots are those of P A P^T (still "pivot <= 0
// <=> not positive definite", log|A| = sum of their logs).  Consequences for the chain of one block step
// (update the next pivot rows -> D^-1 C -> publish):
//   * the owning wave updates ONE register row per lane (32 FMAs; the blocked layout: 64 at half lane occupancy);
//   * the 4 x 4 diagonal block never leaves the quad structure: each lane fetches ITS row of D from the two holder quads
//     with four ds_bpermute (not ten broadcasts of single elements to every lane);
//   * F = D^-1 C comes from four Gauss-Jordan steps on [D | C] inside every quad: the pivot row travels by DPP quad_perm
//     broadcast fused into the FMA, the row scaling and the elimination are ONE uniform instruction stream
//     (coefficient r - 1 on the pivot lane, -d r on the others);
//   * C and F are stored row-major per pivot ([a][128]): the owner writes 2 x 16 bytes per lane and matrix, every reader
//     fetches its 4 consecutive rows of F and 8 columns of C with 16-byte loads (12 per step), where the blocked layout read
//     F with 32 dword loads.
// Everything else is factor.h's scheme: one barrier per block step, the uniform rank-4 update made exact by publishing C
// with D - I at the pivot columns and M_PP := D - 2I, the owner of the next block postponing the rest of its update, raised
// priority on the chain.
//
// What the ablations of tools/history/sweepw_bench.hip say about where a block step (1880 cycles) goes - the reason the layout
// alone bought only 3 %: removing the elimination, the bpermute AND the chain's row update together saves 2 %; the floor of
// "publish -> s_waitcnt -> s_barrier -> wake up" with no arithmetic at all is 690 cycles; the chain's arithmetic adds 640;
// the remaining 550 are the OTHER waves' bulk update (LDS reads + FMAs competing with the chain wave, 300) and the wave
// that pays its postponed update (250).  Hence: the postponed update is spread over the four following steps (one pivot
// per step, NSLOT = 6 slots keep the vectors alive), and the stamp sites double as scheduling fences.
#pragma once
#ifndef ADKF_W_ABLATE
#define ADKF_W_ABLATE 0   // timing-only ablations for tools/history/sweepw_bench.hip (1: no elimination, 2: no bpermute, 4: no chain row update, 8: no postponed update, 16: no bulk update)
#endif

namespace adkf {

template <> struct SweepSmem<128, 512> {
    static constexpr int B = 4;
    static constexpr int NSLOT = 6;          // the postponed update of a step is worked off over the four following steps
    alignas(16) float cross[NSLOT][4][128];  // C: the 4 pivot rows (with D - I at the pivot columns)
    alignas(16) float fvec[NSLOT][4][128];   // F = D^-1 C
    alignas(16) float pivs[128];
    alignas(16) float vec_in[128];
    alignas(16) float vec_out[128];
    float red[8 * 8];
    int redi[8];
#if ADKF_STAMP
    unsigned long long stamp[8 * 16];
#endif
    static constexpr int SCRATCH_FLOATS = NSLOT * 4 * 128;
    __device__ __forceinline__ float* scratch() { return &cross[0][0][0]; }   // free for the caller between two sweeps
};

#if ADKF_STAMP   // diagnostic build (tools/history/sweepw_bench.hip -DADKF_STAMP=<step>): s_memtime of the phases of block step <step>, per wave
#define ADKF_WTS(slot_) do { if (s_stamp == ADKF_STAMP && (threadIdx.x & 63) == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); sm.stamp[(threadIdx.x >> 6) * 16 + (slot_)] = t_; } } while (0)
#elif !defined(ADKF_W_NO_SCHED)
// the stamp sites double as scheduling fences in the product build: left free, hipcc hoists the bulk's LDS loads and sinks the
// publishing stores across the phases of the chain (36.0 us per sweep against 28.1 with the fences, tools/history/sweepw_bench.hip)
#define ADKF_WTS(slot_) __builtin_amdgcn_sched_barrier(0)
#else
#define ADKF_WTS(slot_) do {} while (0)
#endif

template <> struct Sweep<128, 512> {
    using Smem = SweepSmem<128, 512>;
    static constexpr int NMAX = 128, NT = 512, RB = 4, CB = 8, B = 4, NW = 8, NQ = 32;

    __device__ static __forceinline__ int wave() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }
    __device__ static __forceinline__ int bc() { return (threadIdx.x & 63) >> 2; }   // column block
    __device__ static __forceinline__ int row(int r) { return (threadIdx.x >> 6) * 16 + (threadIdx.x & 3) * 4 + r; }
    __device__ static __forceinline__ int col(int c) { return bc() * CB + c; }
    // number of waves whose block of register row g contains a real row (16 w + g < n): blocks beyond are identity padding
    __device__ static __forceinline__ int real_waves(int g, int n) { const int k = (n - g + 15) >> 4; return k < 0 ? 0 : (k > NW ? NW : k); }

    template <int P> __device__ static __forceinline__ float quad_bcast(float v) {
        return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), P * 0x55, 0xF, 0xF, true));
    }

    // rank-4 update of register rows [R0, R1) from the vectors of `slot`
    template <int R0, int R1>
    __device__ static __forceinline__ void apply_rows(float (&m)[RB][CB], int slot, Smem& sm) {
        if constexpr (R0 < R1) {
            const int j0 = bc() * CB, i0 = row(0);
#pragma unroll
            for (int a = 0; a < B; ++a) {
                const float4 f4 = *reinterpret_cast<const float4*>(&sm.fvec[slot][a][i0]);
                const float4 c0 = *reinterpret_cast<const float4*>(&sm.cross[slot][a][j0]);
                const float4 c1 = *reinterpret_cast<const float4*>(&sm.cross[slot][a][j0 + 4]);
                const float fi[4] = {f4.x, f4.y, f4.z, f4.w};
                const float cj[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
                for (int r = R0; r < R1; ++r)
#pragma unroll
                    for (int c = 0; c < CB; ++c) m[r][c] = fmaf(-fi[r], cj[c], m[r][c]);
            }
        }
    }

    // pivot `a` of the rank-4 update of `slot`, applied to every register row but `gx` (the row the chain already brought up
    // to date): the skipped row takes part with a zero factor, so the instruction stream has static register indices whatever
    // gx is (a runtime choice between differently-shaped updates sent the whole block to scratch)
    __device__ static __forceinline__ void apply_pivot_except(float (&m)[RB][CB], int slot, int a, int gx, Smem& sm) {
        const int j0 = bc() * CB, i0 = row(0);
        const float4 f4 = *reinterpret_cast<const float4*>(&sm.fvec[slot][a][i0]);
        const float4 c0 = *reinterpret_cast<const float4*>(&sm.cross[slot][a][j0]);
        const float4 c1 = *reinterpret_cast<const float4*>(&sm.cross[slot][a][j0 + 4]);
        const float fi[4] = {gx == 0 ? 0.f : f4.x, gx == 1 ? 0.f : f4.y, gx == 2 ? 0.f : f4.z, gx == 3 ? 0.f : f4.w};
        const float cj[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int c = 0; c < CB; ++c) m[r][c] = fmaf(-fi[r], cj[c], m[r][c]);
    }

    // Row update of one Gauss-Jordan step inside every quad:  x += g * (x of quad lane P)  for the remaining columns of D and
    // the eight columns of C, as v_fmac_f32 with the DPP quad broadcast on its first source (hipcc does not fold a mov_dpp into
    // the FMA here: it emitted v_mov 0 + v_mov_dpp + v_fmac per element, measured in the .s).  The leading / trailing s_nop
    // cover the "VALU write -> DPP read of the same VGPR" wait states, which the compiler cannot see through an asm statement.
#define ADKF_GJ_LINE(op, P) "v_fmac_f32_dpp " op ", " op ", %[g] quad_perm:[" #P "," #P "," #P "," #P "] row_mask:0xf bank_mask:0xf\n\t"
#define ADKF_GJ_C(P) ADKF_GJ_LINE("%[c0]", P) ADKF_GJ_LINE("%[c1]", P) ADKF_GJ_LINE("%[c2]", P) ADKF_GJ_LINE("%[c3]", P) \
                     ADKF_GJ_LINE("%[c4]", P) ADKF_GJ_LINE("%[c5]", P) ADKF_GJ_LINE("%[c6]", P) ADKF_GJ_LINE("%[c7]", P)
#define ADKF_GJ_COPS(Cr) [c0] "+v"(Cr[0]), [c1] "+v"(Cr[1]), [c2] "+v"(Cr[2]), [c3] "+v"(Cr[3]), [c4] "+v"(Cr[4]), [c5] "+v"(Cr[5]), [c6] "+v"(Cr[6]), [c7] "+v"(Cr[7])
    template <int P>
    __device__ static __forceinline__ void gj_update(float (&D)[4], float (&Cr)[CB], float g) {
        if constexpr (P == 0)
            asm volatile("s_nop 1\n\t" ADKF_GJ_LINE("%[d1]", 0) ADKF_GJ_LINE("%[d2]", 0) ADKF_GJ_LINE("%[d3]", 0) ADKF_GJ_C(0) "s_nop 1"
                         : [d1] "+v"(D[1]), [d2] "+v"(D[2]), [d3] "+v"(D[3]), ADKF_GJ_COPS(Cr) : [g] "v"(g));
        else if constexpr (P == 1)
            asm volatile("s_nop 1\n\t" ADKF_GJ_LINE("%[d2]", 1) ADKF_GJ_LINE("%[d3]", 1) ADKF_GJ_C(1) "s_nop 1"
                         : [d2] "+v"(D[2]), [d3] "+v"(D[3]), ADKF_GJ_COPS(Cr) : [g] "v"(g));
        else if constexpr (P == 2)
            asm volatile("s_nop 1\n\t" ADKF_GJ_LINE("%[d3]", 2) ADKF_GJ_C(2) "s_nop 1"
                         : [d3] "+v"(D[3]), ADKF_GJ_COPS(Cr) : [g] "v"(g));
        else
            asm volatile("s_nop 1\n\t" ADKF_GJ_C(3) "s_nop 1" : ADKF_GJ_COPS(Cr) : [g] "v"(g));
    }

    template <int P>
    __device__ static __forceinline__ void gj_step(float (&D)[4], float (&Cr)[CB], float (&piv)[4], int a) {
        const float dpp = quad_bcast<P>(D[P]);      // the pivot: identical in every lane of the wave
        piv[P] = dpp;
        const float r = fast_rcp(dpp);
        const float g = (a == P) ? (r - 1.f) : (-D[P] * r);   // row P is scaled by r, row a loses d_aP r times row P
        gj_update<P>(D, Cr, g);
    }

    // The owning wave (wn) publishes C and F = D^-1 C of the block whose rows sit in register row G into `slot`.
    template <int G>
    __device__ static __forceinline__ void publish(float (&m)[RB][CB], int wn, int slot, Smem& sm, int s_stamp = -1) {
        (void)s_stamp;
        const int lane = threadIdx.x & 63, a = lane & 3, cb = lane >> 2, j0 = cb * CB;
        // my row of the 4 x 4 diagonal block: columns 16 wn + 4 b + G live in quads 2 wn (b = 0, 1) and 2 wn + 1 (b = 2, 3),
        // registers G (b even) and 4 + G (b odd)
        const int src0 = ((8 * wn + 0) + a) << 2, src1 = ((8 * wn + 4) + a) << 2;   // byte addresses of lanes 4 (2 wn) + a, 4 (2 wn + 1) + a
        float D[4];
#if ADKF_W_ABLATE & 2
        D[0] = m[G][G] + 1.f; D[1] = m[G][4 + G] * 0.01f; D[2] = m[G][1] * 0.01f; D[3] = m[G][2] * 0.01f; (void)src0; (void)src1;
#else
        D[0] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src0, __builtin_bit_cast(int, m[G][G])));
        D[1] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src0, __builtin_bit_cast(int, m[G][4 + G])));
        D[2] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src1, __builtin_bit_cast(int, m[G][G])));
        D[3] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src1, __builtin_bit_cast(int, m[G][4 + G])));
#endif
        ADKF_WTS(3);
        float Cr[CB], piv[4];
#pragma unroll
        for (int c = 0; c < CB; ++c) Cr[c] = m[G][c];
        // the two holder quads: C := D - I at the pivot columns, M_PP := D - 2I  (factor.h, header)
        const int h = cb - 2 * wn;                  // 0 / 1 in the holder quads
        if (h == 0 || h == 1) {
            const float e0 = (a == 2 * h) ? 1.f : 0.f, e1 = (a == 2 * h + 1) ? 1.f : 0.f;
            Cr[G] -= e0; Cr[4 + G] -= e1;
            m[G][G] -= 2.f * e0; m[G][4 + G] -= 2.f * e1;
        }
        *reinterpret_cast<float4*>(&sm.cross[slot][a][j0]) = make_float4(Cr[0], Cr[1], Cr[2], Cr[3]);   // C is final: its stores fly under the elimination
        *reinterpret_cast<float4*>(&sm.cross[slot][a][j0 + 4]) = make_float4(Cr[4], Cr[5], Cr[6], Cr[7]);
        ADKF_WTS(4);
#if ADKF_W_ABLATE & 1
#pragma unroll
        for (int b = 0; b < 4; ++b) piv[b] = fabsf(D[b]) + 1.f;
#pragma unroll
        for (int c = 0; c < CB; ++c) Cr[c] *= 0.001f * D[c & 3];
#else
        gj_step<0>(D, Cr, piv, a);
        gj_step<1>(D, Cr, piv, a);
        ADKF_WTS(5);
        gj_step<2>(D, Cr, piv, a);
        gj_step<3>(D, Cr, piv, a);
#endif
        ADKF_WTS(6);
        *reinterpret_cast<float4*>(&sm.fvec[slot][a][j0]) = make_float4(Cr[0], Cr[1], Cr[2], Cr[3]);
        *reinterpret_cast<float4*>(&sm.fvec[slot][a][j0 + 4]) = make_float4(Cr[4], Cr[5], Cr[6], Cr[7]);
        if (lane == 0) {
#pragma unroll
            for (int b = 0; b < 4; ++b) sm.pivs[16 * wn + 4 * b + G] = piv[b];
        }
        ADKF_WTS(7);
    }

    // The critical path of one block step, run by the wave that owns the NEXT block (register row GN): bring only that
    // row up to date with the vectors of `slot`, publish into `slot_next`; the rest of this step's update is postponed.
    template <int GN>
    __device__ static __forceinline__ void chain(float (&m)[RB][CB], int wn, int slot, int slot_next, Smem& sm, int s_stamp = -1) {
        (void)s_stamp;
        __builtin_amdgcn_s_setprio(3);
        ADKF_WTS(1);
#if !(ADKF_W_ABLATE & 4)
        apply_rows<GN, GN + 1>(m, slot, sm);
#endif
        ADKF_WTS(2);
        publish<GN>(m, wn, slot_next, sm, s_stamp);
        __builtin_amdgcn_s_setprio(0);
    }

    // Postponed work of a wave that ran the chain: the update of step `s` for every row but `g`, worked off ONE PIVOT PER STEP
    // over the following four steps (all at once it made this wave the slowest of its next step).
    struct Owed { int s, g, k; };   // k = next pivot to apply (B: nothing owed)

    __device__ static __forceinline__ void pay_one(float (&m)[RB][CB], Owed& o, Smem& sm) {
#if !(ADKF_W_ABLATE & 8)
        if (o.k < B) { apply_pivot_except(m, o.s % Smem::NSLOT, o.k, o.g, sm); ++o.k; }
#else
        o.k = B;
#endif
    }
    __device__ static __forceinline__ void pay_all(float (&m)[RB][CB], Owed& o, Smem& sm) {
#pragma unroll
        for (int i = 0; i < B; ++i) pay_one(m, o, sm);
    }

    // All block steps whose pivot rows are register row G.  `s` counts executed steps (slot = s mod NSLOT).
    template <int G>
    __device__ static __forceinline__ void phase(float (&m)[RB][CB], int n, int& s, Owed& owed, Smem& sm) {
        if constexpr (G < RB) {
            const int wv = wave();
            const int nw = real_waves(G, n);
            const int nw_next = (G + 1 < RB) ? real_waves(G + 1, n) : 0;
            for (int w = 0; w < nw; ++w) {
                __syncthreads();                       // the vectors of block (G, w) are in slot s mod NSLOT
                const int slot = s % Smem::NSLOT, slot_next = (s + 1) % Smem::NSLOT;
                const bool last = (w + 1 == nw);
                const bool has_next = !last || nw_next > 0;
                const int wn = last ? 0 : w + 1;
                const int s_stamp = s; (void)s_stamp;
                ADKF_WTS(0);
                if (has_next && wv == wn) {
                    if (owed.k < B) pay_all(m, owed, sm);   // (only when one wave owns consecutive blocks: tiny n)
                    if (!last) {
                        chain<G>(m, wn, slot, slot_next, sm, s);
                        owed = Owed{s, G, 0};
                    } else {
                        if constexpr (G + 1 < RB) { chain<G + 1>(m, wn, slot, slot_next, sm, s); owed = Owed{s, G + 1, 0}; }
                    }
                } else {
#if !(ADKF_W_ABLATE & 16)
                    apply_rows<0, RB>(m, slot, sm);
#endif
                    ADKF_WTS(9);
                    pay_one(m, owed, sm);              // slot lifetime: step s is overwritten by the publish of step s + NSLOT - 1 = s + 5
                    ADKF_WTS(8);
                }
                ++s;
            }
            phase<G + 1>(m, n, s, owed, sm);
        }
    }

    // In: m = this thread's block of the SPD matrix (identity-padded beyond n).  Out: m = -(A^-1); the pivots are
    // left in sm.pivs[0..n) (by matrix index).  All threads call.
    __device__ static __forceinline__ void run(float (&m)[RB][CB], int n, Smem& sm) {
        int s = 0;
        Owed owed{0, 0, B};
        if (n > 0 && wave() == 0) publish<0>(m, 0, 0, sm);
        phase<0>(m, n, s, owed, sm);
        pay_all(m, owed, sm);                          // (the owners of the last blocks may still owe up to three pivots)
        __syncthreads();
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

    // out[i] = sum_j (-m_ij) in[j], i.e. A^-1 * in.  `in` must be visible (barrier before); `out` is visible on return.
    __device__ static __forceinline__ void solve(const float (&m)[RB][CB], const float* in, float* out) {
        const int j0 = bc() * CB;
        const float4 x0 = *reinterpret_cast<const float4*>(in + j0), x1 = *reinterpret_cast<const float4*>(in + j0 + 4);
        const float x[CB] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
        float sr[RB];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CB; ++c) acc = fmaf(-m[r][c], x[c], acc);
            sr[r] = acc;
        }
        // the 16 column blocks of a row sit in lanes a, a + 4, .., a + 60
        // (inside a row of 16 lanes: two DPP rotations by 4 and 8; across the four rows: two shuffles)
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            sr[r] += dpp_f<0x124>(sr[r]);   // row_ror:4
            sr[r] += dpp_f<0x128>(sr[r]);   // row_ror:8
        }
#pragma unroll
        for (int o = 16; o < 64; o <<= 1)
#pragma unroll
            for (int r = 0; r < RB; ++r) sr[r] += __shfl_xor(sr[r], o, 64);
        if (bc() == 0) *reinterpret_cast<float4*>(out + row(0)) = make_float4(sr[0], sr[1], sr[2], sr[3]);
        __syncthreads();
    }
};

}  // namespace adkf
