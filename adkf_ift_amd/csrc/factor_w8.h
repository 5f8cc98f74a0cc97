// Sweep<128, 512>, eight pivots per barrier: factor_w.h's wave-owned layout with TWO 4-pivot blocks per block step.
//
// Why: the ablations of tools/sweepw_bench.hip price a block step of factor_w.h at 690 cycles of pure hand-off (publish ->
// s_waitcnt -> s_barrier -> wake-up -> LDS reads, no arithmetic at all) + 640 of chain arithmetic + ~460 of interference from
// the other waves' update.  The hand-off is paid per BARRIER, not per pivot.  Here the owning wave sweeps register rows 2h and
// 2h + 1 (matrix rows 16 w + 4 a + 2h, + 2h + 1) back to back: the second block needs the first block's vectors only for ONE
// register row of the same wave - F through eight ds_bpermute, C straight from the quad lanes by DPP - so no barrier and no
// LDS round trip separates the two eliminations.  16 barriers per sweep instead of 32.
//
// Bookkeeping (all factors of skipped rows are zeroed, so every update has static register indices):
//   * a macro step s publishes 8 pivots (block 0: slots [0,4), block 1: [4,8)) into ring slot (s + 1) mod NSLOT;
//   * the owner brings only its two pivot rows up to date with the previous macro step (128 FMAs) and owes the other two rows:
//     paid two pivots per step over the next four steps (NSLOT = 6 keeps the vectors alive);
//   * inside its own macro step the owner has applied block 0 to its row 2h + 1 only: when it later meets its own slot as a
//     regular wave it applies block 0 to the other three rows and block 1 to all four.
// Layout, elimination (DPP Gauss-Jordan inside the quads), pivots and padding exactly as in factor_w.h.
#pragma once
#ifndef ADKF_W_ABLATE
#define ADKF_W_ABLATE 0
#endif

namespace adkf {

template <> struct SweepSmem<128, 512> {
    static constexpr int B = 4;
    static constexpr int NSLOT = 6;
    alignas(16) float cross[NSLOT][8][128];  // C: the 8 pivot rows of a macro step (with D - I at their pivot columns)
    alignas(16) float fvec[NSLOT][8][128];   // F = D^-1 C, block by block
    alignas(16) float pivs[128];
    alignas(16) float vec_in[128];
    alignas(16) float vec_out[128];
    float red[8 * 8];
    int redi[8];
};

template <> struct Sweep<128, 512> {
    using Smem = SweepSmem<128, 512>;
    static constexpr int NMAX = 128, NT = 512, RB = 4, CB = 8, B = 4, NW = 8;

    __device__ static __forceinline__ int wave() { return __builtin_amdgcn_readfirstlane(threadIdx.x >> 6); }
    __device__ static __forceinline__ int bc() { return (threadIdx.x & 63) >> 2; }
    __device__ static __forceinline__ int row(int r) { return (threadIdx.x >> 6) * 16 + (threadIdx.x & 3) * 4 + r; }
    __device__ static __forceinline__ int col(int c) { return bc() * CB + c; }
    // waves whose macro step h (register rows 2h, 2h + 1) contains a real row: 16 w + 2h < n
    __device__ static __forceinline__ int real_waves(int h, int n) { const int k = (n - 2 * h + 15) >> 4; return k < 0 ? 0 : (k > NW ? NW : k); }

    template <int P> __device__ static __forceinline__ float quad_bcast(float v) {
        return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), P * 0x55, 0xF, 0xF, true));
    }

    // pivots [P0, P1) of ring slot `slot` applied to this thread's block; rows whose bit is set in `skip` take a zero factor
    template <int P0, int P1>
    __device__ static __forceinline__ void apply(float (&m)[RB][CB], int slot, int skip, Smem& sm) {
        const int j0 = bc() * CB, i0 = row(0);
#pragma unroll
        for (int p = P0; p < P1; ++p) {
            const float4 f4 = *reinterpret_cast<const float4*>(&sm.fvec[slot][p][i0]);
            const float4 c0 = *reinterpret_cast<const float4*>(&sm.cross[slot][p][j0]);
            const float4 c1 = *reinterpret_cast<const float4*>(&sm.cross[slot][p][j0 + 4]);
            const float fi[4] = {(skip & 1) ? 0.f : f4.x, (skip & 2) ? 0.f : f4.y, (skip & 4) ? 0.f : f4.z, (skip & 8) ? 0.f : f4.w};
            const float cj[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
            for (int r = 0; r < RB; ++r)
#pragma unroll
                for (int c = 0; c < CB; ++c) m[r][c] = fmaf(-fi[r], cj[c], m[r][c]);
        }
    }
    // the same for ONE pivot chosen at run time (the postponed updates)
    __device__ static __forceinline__ void apply_one(float (&m)[RB][CB], int slot, int p, int skip, Smem& sm) {
        const int j0 = bc() * CB, i0 = row(0);
        const float4 f4 = *reinterpret_cast<const float4*>(&sm.fvec[slot][p][i0]);
        const float4 c0 = *reinterpret_cast<const float4*>(&sm.cross[slot][p][j0]);
        const float4 c1 = *reinterpret_cast<const float4*>(&sm.cross[slot][p][j0 + 4]);
        const float fi[4] = {(skip & 1) ? 0.f : f4.x, (skip & 2) ? 0.f : f4.y, (skip & 4) ? 0.f : f4.z, (skip & 8) ? 0.f : f4.w};
        const float cj[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int c = 0; c < CB; ++c) m[r][c] = fmaf(-fi[r], cj[c], m[r][c]);
    }
    // rows R0 and R0 + 1 only, all 8 pivots (the owner's two pivot rows before its macro step)
    template <int R0>
    __device__ static __forceinline__ void apply_two_rows(float (&m)[RB][CB], int slot, Smem& sm) {
        const int j0 = bc() * CB, i0 = row(0);
#pragma unroll
        for (int p = 0; p < 8; ++p) {
            const float2 f2 = *reinterpret_cast<const float2*>(&sm.fvec[slot][p][i0 + R0]);
            const float4 c0 = *reinterpret_cast<const float4*>(&sm.cross[slot][p][j0]);
            const float4 c1 = *reinterpret_cast<const float4*>(&sm.cross[slot][p][j0 + 4]);
            const float cj[8] = {c0.x, c0.y, c0.z, c0.w, c1.x, c1.y, c1.z, c1.w};
#pragma unroll
            for (int c = 0; c < CB; ++c) { m[R0][c] = fmaf(-f2.x, cj[c], m[R0][c]); m[R0 + 1][c] = fmaf(-f2.y, cj[c], m[R0 + 1][c]); }
        }
    }

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
        const float dpp = quad_bcast<P>(D[P]);
        piv[P] = dpp;
        const float r = fast_rcp(dpp);
        const float g = (a == P) ? (r - 1.f) : (-D[P] * r);
        gj_update<P>(D, Cr, g);
    }

    // x[c] += nf * (x0[c] of quad lane P), eight columns: the owner's own row 2h + 1 takes block 0's update from the quad lanes
#define ADKF_LU_LINE(dst, src, P) "v_fmac_f32_dpp " dst ", " src ", %[g] quad_perm:[" #P "," #P "," #P "," #P "] row_mask:0xf bank_mask:0xf\n\t"
#define ADKF_LU_ALL(P) ADKF_LU_LINE("%[m0]", "%[s0]", P) ADKF_LU_LINE("%[m1]", "%[s1]", P) ADKF_LU_LINE("%[m2]", "%[s2]", P) ADKF_LU_LINE("%[m3]", "%[s3]", P) \
                       ADKF_LU_LINE("%[m4]", "%[s4]", P) ADKF_LU_LINE("%[m5]", "%[s5]", P) ADKF_LU_LINE("%[m6]", "%[s6]", P) ADKF_LU_LINE("%[m7]", "%[s7]", P)
    template <int P>
    __device__ static __forceinline__ void local_update(float (&x)[CB], const float (&x0)[CB], float nf) {
#define ADKF_LU_OPS : [m0] "+v"(x[0]), [m1] "+v"(x[1]), [m2] "+v"(x[2]), [m3] "+v"(x[3]), [m4] "+v"(x[4]), [m5] "+v"(x[5]), [m6] "+v"(x[6]), [m7] "+v"(x[7]) \
                    : [s0] "v"(x0[0]), [s1] "v"(x0[1]), [s2] "v"(x0[2]), [s3] "v"(x0[3]), [s4] "v"(x0[4]), [s5] "v"(x0[5]), [s6] "v"(x0[6]), [s7] "v"(x0[7]), [g] "v"(nf)
        if constexpr (P == 0) asm volatile("s_nop 1\n\t" ADKF_LU_ALL(0) "s_nop 1" ADKF_LU_OPS);
        else if constexpr (P == 1) asm volatile("s_nop 1\n\t" ADKF_LU_ALL(1) "s_nop 1" ADKF_LU_OPS);
        else if constexpr (P == 2) asm volatile("s_nop 1\n\t" ADKF_LU_ALL(2) "s_nop 1" ADKF_LU_OPS);
        else asm volatile("s_nop 1\n\t" ADKF_LU_ALL(3) "s_nop 1" ADKF_LU_OPS);
#undef ADKF_LU_OPS
    }

    // One 4-pivot block of the owning wave wn (register row G): C (with D - I at the pivot columns) and F = D^-1 C go to pivots
    // [P0, P0 + 4) of `slot`; on return Cr0 = this lane's row of C (as published), Cr = its row of F.
    template <int G, int P0>
    __device__ static __forceinline__ void publish(float (&m)[RB][CB], int wn, int slot, float (&Cr0)[CB], float (&Cr)[CB], Smem& sm) {
        const int lane = threadIdx.x & 63, a = lane & 3, cb = lane >> 2, j0 = cb * CB;
        const int src0 = ((8 * wn + 0) + a) << 2, src1 = ((8 * wn + 4) + a) << 2;
        float D[4];
        D[0] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src0, __builtin_bit_cast(int, m[G][G])));
        D[1] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src0, __builtin_bit_cast(int, m[G][4 + G])));
        D[2] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src1, __builtin_bit_cast(int, m[G][G])));
        D[3] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src1, __builtin_bit_cast(int, m[G][4 + G])));
        __builtin_amdgcn_sched_barrier(0);
        float piv[4];
#pragma unroll
        for (int c = 0; c < CB; ++c) Cr0[c] = m[G][c];
        const int h = cb - 2 * wn;
        if (h == 0 || h == 1) {
            const float e0 = (a == 2 * h) ? 1.f : 0.f, e1 = (a == 2 * h + 1) ? 1.f : 0.f;
            Cr0[G] -= e0; Cr0[4 + G] -= e1;
            m[G][G] -= 2.f * e0; m[G][4 + G] -= 2.f * e1;
        }
#pragma unroll
        for (int c = 0; c < CB; ++c) Cr[c] = Cr0[c];
        *reinterpret_cast<float4*>(&sm.cross[slot][P0 + a][j0]) = make_float4(Cr0[0], Cr0[1], Cr0[2], Cr0[3]);
        *reinterpret_cast<float4*>(&sm.cross[slot][P0 + a][j0 + 4]) = make_float4(Cr0[4], Cr0[5], Cr0[6], Cr0[7]);
        __builtin_amdgcn_sched_barrier(0);
        gj_step<0>(D, Cr, piv, a);
        gj_step<1>(D, Cr, piv, a);
        gj_step<2>(D, Cr, piv, a);
        gj_step<3>(D, Cr, piv, a);
        __builtin_amdgcn_sched_barrier(0);
        *reinterpret_cast<float4*>(&sm.fvec[slot][P0 + a][j0]) = make_float4(Cr[0], Cr[1], Cr[2], Cr[3]);
        *reinterpret_cast<float4*>(&sm.fvec[slot][P0 + a][j0 + 4]) = make_float4(Cr[4], Cr[5], Cr[6], Cr[7]);
        if (lane == 0) {
#pragma unroll
            for (int b = 0; b < 4; ++b) sm.pivs[16 * wn + 4 * b + G] = piv[b];
        }
        __builtin_amdgcn_sched_barrier(0);
    }

    // The owner's macro step H (register rows 2H, 2H + 1): block 0, its update of row 2H + 1 inside the wave, block 1.
    template <int H>
    __device__ static __forceinline__ void macro_publish(float (&m)[RB][CB], int wn, int slot, Smem& sm) {
        constexpr int G0 = 2 * H, G1 = 2 * H + 1;
        const int lane = threadIdx.x & 63, a = lane & 3;
        float C0[CB], F0[CB];
        publish<G0, 0>(m, wn, slot, C0, F0, sm);
        // F0[p][my row G1] for p = 0..3: column 16 wn + 4 a + G1 of F row p sits in lane (p, quad 2 wn + (a >> 1)), register
        // 4 (a & 1) + G1 - which of the two registers depends on the READER's a, so both are fetched and one is kept
        float nf[4];
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const int src = (p + 4 * (2 * wn + (a >> 1))) << 2;
            const float lo = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, F0[G1])));
            const float hi = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src, __builtin_bit_cast(int, F0[4 + G1])));
            nf[p] = -((a & 1) ? hi : lo);
        }
        __builtin_amdgcn_sched_barrier(0);
        local_update<0>(m[G1], C0, nf[0]);
        local_update<1>(m[G1], C0, nf[1]);
        local_update<2>(m[G1], C0, nf[2]);
        local_update<3>(m[G1], C0, nf[3]);
        __builtin_amdgcn_sched_barrier(0);
        float C1[CB], F1[CB];
        publish<G1, 4>(m, wn, slot, C1, F1, sm);
    }

    struct Owed { int s, skip, k; };   // the update of macro step s still owed to the rows NOT in `skip`, pivots k..7 (8: nothing owed)

    __device__ static __forceinline__ void pay(float (&m)[RB][CB], Owed& o, int count, Smem& sm) {
#if !(ADKF_W_ABLATE & 8)
        for (int i = 0; i < count; ++i)
            if (o.k < 8) { apply_one(m, o.s % Smem::NSLOT, o.k, o.skip, sm); ++o.k; }
#else
        o.k = 8;
#endif
    }

    template <int H>
    __device__ static __forceinline__ void phase(float (&m)[RB][CB], int n, int& s, Owed& owed, int& own_step, Smem& sm) {
        if constexpr (H < 2) {
            const int wv = wave();
            const int nw = real_waves(H, n);
            const int nw_next = (H + 1 < 2) ? real_waves(H + 1, n) : 0;
            for (int w = 0; w < nw; ++w) {
                __syncthreads();                       // the 8 vectors of macro step (H, w) are in slot s mod NSLOT
                const int slot = s % Smem::NSLOT, slot_next = (s + 1) % Smem::NSLOT;
                const bool last = (w + 1 == nw);
                const bool has_next = !last || nw_next > 0;
                const int wn = last ? 0 : w + 1;
                __builtin_amdgcn_sched_barrier(0);
                if (has_next && wv == wn) {
                    if (owed.k < 8) pay(m, owed, 8, sm);   // (only when one wave owns consecutive macro steps: tiny n)
                    __builtin_amdgcn_s_setprio(3);
                    if (s == own_step) {                    // (tiny n again: this wave also published the slot it now consumes)
                        apply<0, 4>(m, slot, 2 << (2 * H), sm);
                        apply<4, 8>(m, slot, 0, sm);
                        if (!last) { macro_publish<H>(m, wn, slot_next, sm); }
                        else { if constexpr (H + 1 < 2) macro_publish<H + 1>(m, wn, slot_next, sm); }
                        owed.k = 8;
                    } else {
                        if (!last) {
                            apply_two_rows<2 * H>(m, slot, sm);
                            __builtin_amdgcn_sched_barrier(0);
                            macro_publish<H>(m, wn, slot_next, sm);
                            owed = Owed{s, 3 << (2 * H), 0};
                        } else {
                            if constexpr (H + 1 < 2) {
                                apply_two_rows<2 * (H + 1)>(m, slot, sm);
                                __builtin_amdgcn_sched_barrier(0);
                                macro_publish<H + 1>(m, wn, slot_next, sm);
                                owed = Owed{s, 3 << (2 * (H + 1)), 0};
                            }
                        }
                    }
                    __builtin_amdgcn_s_setprio(0);
                    own_step = s + 1;
                } else {
#if !(ADKF_W_ABLATE & 16)
                    if (s == own_step) {               // my own macro step: block 0 already reached my row 2H + 1 inside the wave
                        apply<0, 4>(m, slot, 2 << (2 * H), sm);
                        apply<4, 8>(m, slot, 0, sm);
                    } else {
                        apply<0, 8>(m, slot, 0, sm);
                    }
#endif
                    __builtin_amdgcn_sched_barrier(0);
                    pay(m, owed, 2, sm);               // slot lifetime: step s is overwritten by the publish of step s + NSLOT - 1 = s + 5
                }
                __builtin_amdgcn_sched_barrier(0);
                ++s;
            }
            phase<H + 1>(m, n, s, owed, own_step, sm);
        }
    }

    __device__ static __forceinline__ void run(float (&m)[RB][CB], int n, Smem& sm) {
        int s = 0, own_step = -1;
        Owed owed{0, 0, 8};
        if (n > 0 && wave() == 0) { macro_publish<0>(m, 0, 0, sm); own_step = 0; }
        phase<0>(m, n, s, owed, own_step, sm);
        pay(m, owed, 8, sm);
        __syncthreads();
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
#pragma unroll
        for (int o = 4; o < 64; o <<= 1)
#pragma unroll
            for (int r = 0; r < RB; ++r) sr[r] += __shfl_xor(sr[r], o, 64);
        if (bc() == 0) *reinterpret_cast<float4*>(out + row(0)) = make_float4(sr[0], sr[1], sr[2], sr[3]);
        __syncthreads();
    }
};

}  // namespace adkf
