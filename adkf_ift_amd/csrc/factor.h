// Workgroup-cooperative factorisation + inversion of one SPD matrix (n <= NMAX) with the matrix held in
// REGISTERS: thread (br, bc) owns the RB x CB block  rows br*RB.., cols bc*CB..  of the full symmetric matrix.
//
// Algorithm: the symmetric sweep operator (Gauss-Jordan without pivoting), blocked by B = CB pivots.  The pivots
// p_k are exactly those of the LDL^T (the squared Cholesky diagonal), so "p_k <= 0" is the reference's
// not-positive-definite condition and log|A| = sum_k log p_k; after sweeping every index M = -A^-1.
//
// One block step sweeps the B x B diagonal block D = M_PP (held by ONE thread).  With C = M_P. (the B pivot rows)
// and F = D^-1 C the classical block sweep is
//       M_RR -= C_R^T F_R,   M_RP = F_R^T,   M_PR = F_R,   M_PP = -D^-1.
// All four cases collapse into ONE uniform rank-B update  M_ij -= sum_a F_a,i C_a,j  over the whole matrix - no
// per-element select in the hot loop - if the holder of D first replaces, in its own registers, the C entries at
// the pivot columns by D - I and M_PP by D - 2I (F = D^-1 C then gives I - D^-1 there by itself):
//       i in R, j = P_b :  c_b,i - sum_a f_a,i (D_ab - d_ab)                     = f_b,i
//       i = P_b, j in R :  c_b,j - sum_a (d_ab - Dinv_ab) c_a,j                  = f_b,j
//       i = P_b, j = P_c:  (D_bc - 2 d_bc) - sum_a (d_ab - Dinv_ab)(D_ac - d_ac) = -Dinv_bc
// Per block step only the B pivot rows (C) and their scaled copies (F) travel through LDS, published by the 1/NBR
// of the threads that own them, with ONE barrier per B pivots; everything else is register FMAs.
// The owners obtain D by v_readlane from the holder's lane and invert it redundantly (same instruction stream).
// Rows/cols >= n are padded with the identity; sweeping them is a no-op that leaves -1 on the diagonal.
//
// Scheduling.  The chain  update(next pivot rows) -> D^-1 -> F -> LDS  is the critical path (every block step
// waits for it; measured ~0.4 us against ~0.23 us of bulk FMAs per step, tools/history/chain_bench.hip), so:
//   * a thread's RB rows are G = RB/CB groups of B rows that lie NMAX/G apart, and consecutive logical block
//     rows live in DIFFERENT waves: the ownership of the chain rotates over the waves from step to step;
//   * (measured and rejected: sending the NEXT diagonal block from its holder before step q's update and letting every
//     lane of the block row update a private copy - 10 x 4 FMAs - instead of waiting for the post-update broadcast:
//     27.4 -> 32.7 us per sweep, the extra scalar FMAs and vector reads cost more than the ~450 cycles they hide;
//     nor a store / read-back of the block through LDS inside the owning wave instead of the ten ds_bpermute: 29.2 us;
//     nor publishing C and D^-1 only and letting every consumer form F = D^-1 C for its own rows: 40.9 us)
//   * the owning wave updates only the next pivot rows, runs the chain at raised priority and goes straight to
//     the barrier; it applies the REST of that step's update one step later, when another wave is on the chain
//     (vector slots are triple-buffered so the old vectors are still there).
#pragma once
#include <limits.h>
#ifndef ADKF_STAMP
#define ADKF_STAMP 0  // diagnostic build only (tools/history/sweep_bench.hip): s_memtime stamps of one block step into sm.stamp[]
#endif
#if ADKF_STAMP
#define ADKF_TS(slot) do { if (q_stamp == ADKF_STAMP && (threadIdx.x & 63) == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) :: "memory"); sm.stamp[(threadIdx.x >> 6) * 16 + (slot)] = t_; } } while (0)
#else
#define ADKF_TS(slot) do {} while (0)
#endif
#ifndef ADKF_ABLATE
#define ADKF_ABLATE 0  // timing-only ablation switches for tools/history/sweep_bench.hip (1: no inverse, 2: no readlane, 4: no F, 8: no deferral)
#endif

#include "device_utils.h"

namespace adkf {

template <int NMAX, int NT> struct SweepCfg;
#ifndef ADKF_CFG128_RB
#define ADKF_CFG128_RB 8
#define ADKF_CFG128_CB 4
#endif
template <> struct SweepCfg<128, 512> { static constexpr int RB = ADKF_CFG128_RB, CB = ADKF_CFG128_CB; };
template <> struct SweepCfg<64, 256> { static constexpr int RB = 4, CB = 4; };
template <> struct SweepCfg<32, 256> { static constexpr int RB = 2, CB = 2; };
template <> struct SweepCfg<16, 256> { static constexpr int RB = 1, CB = 1; };

template <int NMAX, int NT>
struct SweepSmemBlk {
    static constexpr int B = SweepCfg<NMAX, NT>::CB;
    alignas(16) float cross[3][B][NMAX];  // C: the B pivot rows (with D - I at the pivot columns); 3 slots, see steps()
    alignas(16) float fvec[3][B][NMAX];   // F = D^-1 C
    alignas(16) float pivs[NMAX];
    alignas(16) float vec_in[NMAX];       // right-hand side of the solve (y or r)
    alignas(16) float vec_out[NMAX];      // A^-1 * vec_in
    float red[8 * (NT / 64)];
    int redi[NT / 64];
#if ADKF_STAMP
    unsigned long long stamp[(NT / 64) * 16];
#endif
    static constexpr int SCRATCH_FLOATS = 3 * B * NMAX;
    __device__ __forceinline__ float* scratch() { return &cross[0][0][0]; }   // free for the caller between two sweeps
};

__device__ __forceinline__ float fast_rcp(float p) {
    float r = __builtin_amdgcn_rcpf(p);     // v_rcp_f32, 1 ulp
    return fmaf(fmaf(-p, r, 1.f), r, r);    // one Newton step: ~0.5 ulp, 3 instructions instead of the 10 of a full divide
}

// In-register inverse of the B x B SPD pivot block; piv = its successive LDL^T pivots.  This sits on the critical
// path of the whole sweep (every block step waits for it), so it is written for instruction-level parallelism:
// closed-form 2 x 2 inverses and one Schur complement - two dependent reciprocals instead of four.
__device__ __forceinline__ void inv2(float a, float b, float c, float& ia, float& ib, float& ic, float& det) {
    det = fmaf(a, c, -b * b);
#ifdef ADKF_RAW_RCP
    const float r = __builtin_amdgcn_rcpf(det);
#else
    const float r = fast_rcp(det);
#endif
    ia = c * r; ib = -b * r; ic = a * r;
}
template <int B> struct InvSpd;
template <> struct InvSpd<1> {
    __device__ static __forceinline__ void run(float (&D)[1][1], float (&piv)[1]) { piv[0] = D[0][0]; D[0][0] = fast_rcp(D[0][0]); }
};
template <> struct InvSpd<2> {
    __device__ static __forceinline__ void run(float (&D)[2][2], float (&piv)[2]) {
        float ia, ib, ic, det;
        const float a = D[0][0];
        inv2(a, D[0][1], D[1][1], ia, ib, ic, det);
        piv[0] = a; piv[1] = det * fast_rcp(a);
        D[0][0] = ia; D[0][1] = ib; D[1][0] = ib; D[1][1] = ic;
    }
};
template <> struct InvSpd<4> {
    __device__ static __forceinline__ void run(float (&D)[4][4], float (&piv)[4]) {
        // D = [[P, Q], [Q^T, R]]
        float pa, pb, pc, detP;
        inv2(D[0][0], D[0][1], D[1][1], pa, pb, pc, detP);                 // P^-1
        const float q00 = D[0][2], q01 = D[0][3], q10 = D[1][2], q11 = D[1][3];
        const float t00 = fmaf(pa, q00, pb * q10), t01 = fmaf(pa, q01, pb * q11);   // T = P^-1 Q
        const float t10 = fmaf(pb, q00, pc * q10), t11 = fmaf(pb, q01, pc * q11);
        const float s00 = D[2][2] - fmaf(q00, t00, q10 * t10);                        // S = R - Q^T T
        const float s01 = D[2][3] - fmaf(q00, t01, q10 * t11);
        const float s11 = D[3][3] - fmaf(q01, t01, q11 * t11);
        float sa, sb, sc, detS;
        inv2(s00, s01, s11, sa, sb, sc, detS);                             // S^-1
        const float u00 = fmaf(t00, sa, t01 * sb), u01 = fmaf(t00, sb, t01 * sc);   // U = T S^-1
        const float u10 = fmaf(t10, sa, t11 * sb), u11 = fmaf(t10, sb, t11 * sc);
        piv[0] = D[0][0]; piv[1] = detP * fast_rcp(D[0][0]); piv[2] = s00; piv[3] = detS * fast_rcp(s00);
        D[0][0] = pa + fmaf(u00, t00, u01 * t01); D[0][1] = pb + fmaf(u00, t10, u01 * t11); D[1][1] = pc + fmaf(u10, t10, u11 * t11);
        D[1][0] = D[0][1];
        D[0][2] = -u00; D[0][3] = -u01; D[1][2] = -u10; D[1][3] = -u11;
        D[2][0] = -u00; D[3][0] = -u01; D[2][1] = -u10; D[3][1] = -u11;
        D[2][2] = sa; D[2][3] = sb; D[3][2] = sb; D[3][3] = sc;
    }
};

template <int NMAX, int NT>
struct SweepBlk {
    using Smem = SweepSmemBlk<NMAX, NT>;
    static constexpr int RB = SweepCfg<NMAX, NT>::RB, CB = SweepCfg<NMAX, NT>::CB, B = CB;
    static constexpr int NBC = NMAX / CB, NBR = NMAX / RB;
    static constexpr int G = RB / CB;          // row groups per thread
    static constexpr int GSTRIDE = NMAX / G;   // distance between the row groups (= NBR * CB)
    static constexpr int NW = NT / 64;         // waves
    static constexpr int BPW = 64 / NBC;       // physical block rows per wave
    static_assert(NBC * NBR == NT, "one block per thread");
    static_assert(NBC <= 64 && 64 % NBC == 0, "a row of blocks must sit inside one wave (shuffle reduction, pivot broadcast)");
    static_assert(RB % CB == 0, "CB must divide RB");
    static_assert(NBR == NW * BPW, "block rows must tile the waves");

    __device__ static __forceinline__ int bc() { return threadIdx.x % NBC; }
    // logical block row: physical block row (wave * BPW + h) holds logical row (wave + NW * h), so that logical
    // rows bl, bl + 1, ... sit in waves bl % NW, (bl + 1) % NW, ...
    __device__ static __forceinline__ int br() {
        const int brr = threadIdx.x / NBC;
        return (brr / BPW) + NW * (brr % BPW);
    }
    __device__ static __forceinline__ int row(int r) { return (r / CB) * GSTRIDE + br() * CB + (r % CB); }
    __device__ static __forceinline__ int col(int c) { return bc() * CB + c; }
    __device__ static __forceinline__ int owner_wave(int bl) { return bl % NW; }

    // The owners of pivot block q = GI * NBR + bl (matrix rows q*B .. q*B+B-1 = local rows GI*CB.. of logical block
    // row bl) publish C and F for block step q into `slot`.  GI is a compile-time constant: static register indices.
    template <int GI>
    __device__ static __forceinline__ void publish(float (&m)[RB][CB], int bl, int slot, Smem& sm) {
        constexpr int RO = GI * CB;
        const int q = GI * NBR + bl;
        const int q_stamp = q - 1; (void)q_stamp;
        const int plane = ((bl / NW) * NBC + q) & 63;  // lane, in the owning wave, of the thread (bl, bc = q) that holds D
        float D[B][B];
#pragma unroll
        for (int a = 0; a < B; ++a)
#pragma unroll
            for (int b = 0; b <= a; ++b) {
#if ADKF_ABLATE & 2
                D[a][b] = m[RO + a][b] + (a == b ? 1.f : 0.f);
#elif ADKF_ABLATE & 128
                D[a][b] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, m[RO + a][b]), plane));
#else
                // ds_bpermute: the ten transfers pipeline through the LDS crossbar behind one wait; ten v_readlane
                // (SGPR round trips with their hazard waits) measured ~390 cycles on this critical path
                D[a][b] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(plane << 2, __builtin_bit_cast(int, m[RO + a][b])));
#endif
                D[b][a] = D[a][b];
            }
        ADKF_TS(3);
        // private copies of C and F (registers are plentiful: at most 32 matrix elements per lane)
        if (br() == bl) {
            const int j0 = bc() * CB;
            float C[B][CB], F[B][CB], piv[B];
#pragma unroll
            for (int a = 0; a < B; ++a)
#pragma unroll
                for (int c = 0; c < CB; ++c) C[a][c] = m[RO + a][c];
            if (bc() == q) {  // this thread holds D: C := D - I at the pivot columns, M_PP := D - 2I
#pragma unroll
                for (int a = 0; a < B; ++a) {
                    C[a][a] -= 1.f;
                    m[RO + a][a] -= 2.f;
                }
            }
#pragma unroll
            for (int a = 0; a < B; ++a)
#pragma unroll
                for (int c = 0; c < CB; ++c) sm.cross[slot][a][j0 + c] = C[a][c];  // C is final: its stores fly under the inverse
#if ADKF_ABLATE & 1
            for (int a = 0; a < B; ++a) piv[a] = D[a][a];
#else
            InvSpd<B>::run(D, piv);
#endif
            ADKF_TS(4);
            if (bc() == q) {
#pragma unroll
                for (int a = 0; a < B; ++a) sm.pivs[q * B + a] = piv[a];
            }
#pragma unroll
            for (int a = 0; a < B; ++a)
#pragma unroll
                for (int c = 0; c < CB; ++c) {
#if ADKF_ABLATE & 4
                    F[a][c] = C[a][c] * D[a][a];
#else
                    float s = 0.f;
#pragma unroll
                    for (int b = 0; b < B; ++b) s = fmaf(D[a][b], C[b][c], s);
                    F[a][c] = s;
#endif
                }
#pragma unroll
            for (int a = 0; a < B; ++a)
#pragma unroll
                for (int c = 0; c < CB; ++c) sm.fvec[slot][a][j0 + c] = F[a][c];
            ADKF_TS(6);
        }
    }

    // rank-B update of local rows [R0, R1) of this thread's block from the vectors of block step `slot`
    template <int R0, int R1>
    __device__ static __forceinline__ void apply_pivot(float (&m)[RB][CB], int slot, int a, Smem& sm) {
        const int j0 = bc() * CB;
        float fi[R1 - R0], cj[CB];
#pragma unroll
        for (int r = R0; r < R1; ++r) fi[r - R0] = sm.fvec[slot][a][row(r)];
#pragma unroll
        for (int c = 0; c < CB; ++c) cj[c] = sm.cross[slot][a][j0 + c];
#pragma unroll
        for (int r = R0; r < R1; ++r)
#pragma unroll
            for (int c = 0; c < CB; ++c) m[r][c] = fmaf(-fi[r - R0], cj[c], m[r][c]);
    }
    template <int R0, int R1>
    __device__ static __forceinline__ void apply_step(float (&m)[RB][CB], int slot, Smem& sm) {
        if constexpr (R0 < R1) {
#pragma unroll
            for (int a = 0; a < B; ++a) apply_pivot<R0, R1>(m, slot, a, sm);
        }
    }
    // (for the ablation harness tools/history/sweep_bench.hip)
    __device__ static __forceinline__ void step(float (&m)[RB][CB], int q, Smem& sm) {
        apply_step<0, RB>(m, q % 3, sm);
    }

    // The critical path of one block step, run by the wave that owns the NEXT pivot block qn = NGI * NBR + nbl:
    // fetch only what the chain needs from step q's vectors (C for its columns, F for the next pivot rows), bring
    // those rows up to date, invert and publish - all at raised priority.  The rest of this wave's step-q update
    // happens one step later (see phase()).
    template <int NGI>
    __device__ static __forceinline__ void chain(float (&m)[RB][CB], int nbl, int q, Smem& sm) {
        constexpr int N0 = NGI * CB, N1 = NGI * CB + CB;
        const int q_stamp = q; (void)q_stamp;
        ADKF_TS(0);
        if (!(ADKF_ABLATE & 32)) __builtin_amdgcn_s_setprio(3);
        ADKF_TS(1);
        apply_step<N0, N1>(m, q % 3, sm);
        ADKF_TS(2);
        publish<NGI>(m, nbl, (q + 1) % 3, sm);
        ADKF_TS(7);
        __builtin_amdgcn_s_setprio(0);
    }

    // All block steps whose pivot rows are local row group GI.
    template <int GI>
    __device__ static __forceinline__ void phase(float (&m)[RB][CB], int nq, Smem& sm) {
        if constexpr (GI < G) {
            const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
            for (int bl = 0; bl < NBR; ++bl) {
                const int q = GI * NBR + bl;
                if (q >= nq) break;
                __syncthreads();
                const bool last_of_group = bl + 1 == NBR;
                const bool has_next = q + 1 < nq && (!last_of_group || GI + 1 < G);
                if (has_next && wave == owner_wave(last_of_group ? 0 : bl + 1)) {
                    if (!last_of_group) chain<GI>(m, bl + 1, q, sm);
                    else if constexpr (GI + 1 < G) chain<GI + 1>(m, 0, q, sm);
                } else {
                    const int q_stamp = q; (void)q_stamp;
                    ADKF_TS(0);
#ifdef ADKF_BULK_SLEEP
                    __builtin_amdgcn_s_sleep(ADKF_BULK_SLEEP);  // let the chain owner's LDS reads and issue slots go first
#endif
                    if (q > 0 && wave == owner_wave(bl)) {
                        // this wave ran the chain for block q during the previous step and postponed the rest of that
                        // step's update (everything but its pivot rows GI*CB..): do it now, off the critical path
                        apply_step<0, GI * CB>(m, (q + 2) % 3, sm);  // slot of step q - 1
                        apply_step<GI * CB + CB, RB>(m, (q + 2) % 3, sm);
                    }
                    ADKF_TS(8);
                    apply_step<0, RB>(m, q % 3, sm);
                    ADKF_TS(10);
                }
            }
            phase<GI + 1>(m, nq, sm);
        }
    }

    // In: m = this thread's block of the SPD matrix (identity-padded beyond n).  Out: m = -(A^-1); the pivots are
    // left in sm.pivs[0..n) (finish() turns them into info and log-determinant).  All threads call.
    __device__ static __forceinline__ void run(float (&m)[RB][CB], int n, Smem& sm) {
        const int nq = (n + B - 1) / B;  // pivot blocks that contain a real row; sweeping identity padding is a no-op
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
        if (wave == owner_wave(0)) publish<0>(m, 0, 0, sm);
        phase<0>(m, nq, sm);
        __syncthreads();
        // Opaque re-definition of the block: without it the pairing (SLP) choices of whatever consumes m next leak back
        // into the sweep's register assignment (~190 extra v_mov per two block steps, +34 % sweep time, measured in k_inner).
#pragma unroll
        for (int r = 0; r < RB; ++r)
#pragma unroll
            for (int c = 0; c < CB; ++c) asm volatile("" : "+v"(m[r][c]));
    }

    // log-determinant and info from the pivots.  Contains barriers; all threads call; all get the same values.
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

    // out[i] = sum_j (-m_ij) in[j]  for i < NMAX, i.e. A^-1 * in.  `in` must be visible (barrier before);
    // `out` is visible to all threads on return (barrier inside).
    __device__ static __forceinline__ void solve(const float (&m)[RB][CB], const float* in, float* out) {
        const int j0 = bc() * CB;
        float s[RB];
        float x[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c) x[c] = in[j0 + c];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            float a = 0.f;
#pragma unroll
            for (int c = 0; c < CB; ++c) a -= m[r][c] * x[c];
            s[r] = a;
        }
#pragma unroll
        for (int o = NBC / 2; o > 0; o >>= 1)
#pragma unroll
            for (int r = 0; r < RB; ++r) s[r] += __shfl_xor(s[r], o, 64);
        if (bc() == 0) {
#pragma unroll
            for (int r = 0; r < RB; ++r) out[row(r)] = s[r];
        }
        __syncthreads();
    }
};

// The names the kernels use.  128 points x 512 threads takes the wave-owned variant of factor_w.h (ADKF_SWEEP_W=0 keeps
// the blocked one for A/B measurements); the smaller sizes use the blocked sweep above.
template <int NMAX, int NT> struct SweepSmem : SweepSmemBlk<NMAX, NT> {};
template <int NMAX, int NT> struct Sweep : SweepBlk<NMAX, NT> {};

}  // namespace adkf

// 128 points x 512 threads: rank-4 updates on the matrix pipe (factor_m.h) - the only variant the shipped library compiles.  The
// A/B builds of tools/ (-DADKF_SWEEP_M=0: the VALU variant, -DADKF_SWEEP_M=0 -DADKF_SWEEP_W=0: the blocked sweep above at 128
// points, -DADKF_SWEEP_M=2: the sixteen-pivot experiment, measured slower) take their sweeps from tools/variants/.
#ifndef ADKF_SWEEP_M
#define ADKF_SWEEP_M 1
#endif
#ifndef ADKF_SWEEP_W
#define ADKF_SWEEP_W 1
#endif
#if ADKF_SWEEP_M == 2
#include "../../tools/variants/factor_m16.h"   // A/B experiment only (measured slower); not part of the shipped library
#elif ADKF_SWEEP_M
#include "factor_m.h"
#elif ADKF_SWEEP_W
#include "../../tools/variants/factor_w.h"     // A/B experiment only (the VALU sweep of round 2)
#endif
