// Blocked sweep beyond 128 points, round 5: the update of block step k and the diagonal sweep of block step k + 1 in ONE launch.
//
// large.h runs a block step as three launches - k_lg_diag (one workgroup per task: 23.7 us at C5, of which 17.4 us are the hand-off
// chain of the 128-pivot sweep), the panel product (7.3 us) and the rank-128 update (20.3 us) - strictly one after the other, and
// while a task's ONE sweeping workgroup runs, the other 31 CUs of its XCD idle: a third of the C5 step (profiles/r04_c5_kernel_stats.csv).
// The only thing the sweep of step k + 1 waits for is the diagonal block (k + 1, k + 1) after update k: three 64 x 64 tiles.  Here
// the update launch works through its tiles with THOSE three first; the workgroup whose tile completes the block (an arrival counter
// per task, agent-scope atomic, told by the value its own add returned: no workgroup ever waits or spins) sweeps it at once, while the
// other workgroups of the launch are still updating the remaining tiles.  A block step is then panel + max(update, tiles + sweep)
// instead of sweep + panel + update.
//
// Shape of the launch: 512-thread workgroups (what the sweep of factor_m.h needs), each half (four waves) updating one 64 x 64 tile
// with the arithmetic of k_bgemm<ProbLgUpdate> - same operand staging, same MFMA order, same epilogue expressions: the results are
// BIT-IDENTICAL to the three-launch path (tests/test_gpu_parity.py::test_fused_block_step_equals_three_launches), which stays
// available (ADKF_LG_FUSED=0) for A/B runs and takes the batches whose leading dimension is not a multiple of four.
//
// Visibility (MI355X: a CU's L1 is never refreshed by other CUs' stores, the XCDs' L2s are not coherent): the three tiles of the next
// diagonal block are written WRITE-THROUGH (sc1) by whoever computes them, every storing wave drains its stores (s_waitcnt vmcnt(0)),
// the workgroup's barrier, then ONE lane adds to the counter (agent scope); the workgroup whose add completes the count makes one
// agent-scope acquire (its L1 drops its stale lines), waits for it, barrier, plain loads.  D^-1 is double-buffered by the parity of
// the block step: the sweep of step k + 1 writes the other buffer while the pivot-block tiles of update k still read this one.
#pragma once
#include "large.h"

namespace adkf {

struct LgStepArgs {
    LgMat m;            // m.Dinv: D^-1 of THIS block step
    float* Dinv_next;   // [T, LB, LB] where the sweep of block step + 1 leaves its inverse (the other parity)
    int32_t* cnt;       // [T] arrivals at the next diagonal block; zero between launches (the sweeping workgroup resets it)
    int step, tn, npair, look;   // tn = ceil(ld / 64) tiles per edge, npair = workgroups per task, look = 1: sweep block step + 1 in this launch
    int stagger;        // look = 1: the workgroups WITHOUT a tile of the next diagonal block start this many s_sleep(127) (~3.4 us each) late
    int prio;           // look = 1: wave priority of the sweeping workgroup (s_setprio)
};

constexpr int LGF_NT = 512;
#ifndef ADKF_LGF_EXP
#define ADKF_LGF_EXP 0   // timing experiments of tools/lgf_bench.hip ONLY (results are wrong or unordered): 1 plain stores for the handed-over tiles, 2 no acquire, 4 workgroup 0 sweeps at once (no tiles, no hand-off)
#endif
#ifndef ADKF_LGF_WPS
#define ADKF_LGF_WPS 4   // waves per SIMD the register budget allows: 4 = two workgroups per CU (tools/lgf_bench.hip measures 2 = one per CU as well)
#endif
constexpr int LGF_STAGGER = 0, LGF_PRIO = 0;   // what the library launches with: neither a late start of the other workgroups nor a raised priority of the sweeping one changed the launch time (tools/lgf_bench.hip, profiles/r05_lgf_bench.txt)
constexpr int LGF_LDK = GT + 16;   // [k][mn] operand images of gemm.h (both operands of the update are MN-contiguous)

// relaxed agent-scope stores: global_store_dword ... sc1 (write-through; MI355X_MICROARCH.md, fence table)
__device__ __forceinline__ void st_sc1(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// natural index of tile (ti, tj), ti <= tj, in the row-major enumeration of the upper triangle of a tn x tn tile grid
__device__ __forceinline__ int lgf_tri_index(int tn, int ti, int tj) { return ti * tn - ti * (ti - 1) / 2 + (tj - ti); }
__device__ __forceinline__ void lgf_tri_tile(int tn, int v, int& ti, int& tj) {
    int r = (int)(((float)(2 * tn + 1) - sqrtf((float)((2 * tn + 1) * (2 * tn + 1) - 8 * v))) * 0.5f);
    r = max(0, min(r, tn - 1));
    while (r > 0 && r * tn - r * (r - 1) / 2 > v) --r;
    while (r + 1 < tn && (r + 1) * tn - (r + 1) * r / 2 <= v) ++r;
    ti = r; tj = r + (v - (r * tn - r * (r - 1) / 2));
}

// workgroups per task: three for the tiles of the next diagonal block (one each), the other tiles in pairs
inline __host__ __device__ int lgf_npair(int tn) { return 3 + (tn * (tn + 1) / 2 + 1) / 2; }

__global__ __launch_bounds__(LGF_NT, ADKF_LGF_WPS) void k_lg_update_sweep(LgStepArgs a) {
    using SW = Sweep<128, 512>;
    constexpr int RB = SW::RB, CB = SW::CB;
    __shared__ SweepSmem<128, 512> sm;
    __shared__ float As[2][GK * LGF_LDK];
    __shared__ float Bs[2][GK * LGF_LDK];
    __shared__ int s_last;
    int t, pair;
    if (!task_tile(a.m.T, a.npair, t, pair)) return;
    if (!a.m.active(t)) return;
    const int n = a.m.n(t), ld = a.m.ld, p0 = a.step * LB;
    const int nloc = min(LB, n - p0);
    if (nloc <= 0) return;                       // (task-uniform: every workgroup of the task leaves)
    const int tid = threadIdx.x, h = tid >> 8, lt = tid & 255, lane = tid & 63, wv = lt >> 6;
    const int wr = wv >> 1, wc = wv & 1, fi = lane & 15, fk = lane >> 4;
    float* Mi = a.m.M + (size_t)t * ld * ld;
    const float* Dv = a.m.Dinv + (size_t)t * LB * LB;
    const float* Cb = a.m.Cbuf + (size_t)t * LB * ld;
    const float* Fb = a.m.Fbuf + (size_t)t * LB * ld;
    const int tn = a.tn, ntri = tn * (tn + 1) / 2;
    // ---- which tile: the (up to three) tiles of the next diagonal block first, then the upper triangle in row-major order ----
    const int d0 = 2 * (a.step + 1), d1 = d0 + 1;
    const int nd = (a.look && d0 < tn) ? (d1 < tn ? 3 : 1) : 0;
    // workgroups [0, nd): ONE tile of the next diagonal block each, on the first half - the second half only keeps the barriers company,
    // so that the tile's four waves have the CU's four matrix pipes to themselves (two tiles per workgroup share them: 3.4 us of
    // MFMA issue per tile instead of 1.7, on the critical path of the launch); workgroups [nd, npair): two tiles of the rest each
    int u = -1;
    bool next_diag = false;
    if (pair < nd) { if (h == 0) { u = pair; next_diag = true; } }
    else if (2 * (pair - nd) + h < ntri - nd) u = nd + 2 * (pair - nd) + h;
    int ti = 0, tj = 0;
    bool valid = u >= 0;
    if (valid) {
        if (next_diag) { ti = u < 2 ? d0 : d1; tj = u < 1 ? d0 : d1; }
        else {
            int v = u - nd;
            if (nd >= 1) {
                const int s0 = lgf_tri_index(tn, d0, d0);
                if (v >= s0) ++v;
                if (nd == 3) { if (v >= s0 + 1) ++v; if (v >= lgf_tri_index(tn, d1, d1)) ++v; }
            }
            lgf_tri_tile(tn, v, ti, tj);
        }
    }
    const int m0 = ti * GT, n0 = tj * GT;
    if (m0 >= n || n0 >= n) valid = false;       // tile outside this (ragged) task
#if (ADKF_LGF_EXP & 4)
    const bool direct = nd > 0 && pair == 0;
    if (next_diag) valid = false;
#endif
    // The tiles of the next diagonal block are the head of the launch's critical path (tiles -> sweep: 20 us of hand-off chain), the
    // other tiles have slack.  Started together, the three tiles take as long as a tile of a full launch takes (15 us: every CU of the
    // XCD is loading operands); started alone they take 5.  So everybody else sleeps first.
    if (nd > 0 && pair >= nd)
        for (int q = 0; q < a.stagger; ++q) __builtin_amdgcn_s_sleep(127);
    auto in_p = [&](int i) { return i >= p0 && i < p0 + LB; };
    const bool piv_i = in_p(m0), piv_j = in_p(n0);
    const bool compute = valid && !piv_i && !piv_j;
    // mode (uniform over the four waves of a half): 0 nothing to multiply, 1 every operand load in flight before the first MFMA (full
    // tile of a full block step: gemm.h's DEEP path), 2 range-checked loads chunk by chunk
    const int mode = !compute ? 0 : ((a.m.vec && m0 + GT <= n && n0 + GT <= n && nloc == LB) ? 1 : 2);
    const int kc = (nloc + GK - 1) / GK;         // task-uniform: both halves make the same number of barriers

    f32x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    float* Ah = As[h];
    float* Bh = Bs[h];
    // staging map of gemm.h for MN-contiguous operands: thread -> (k row lt / 16 + 16 ps, four consecutive mn at 4 (lt % 16))
    const int sg = (lt & 15) * 4, sk = lt >> 4;
    auto stage4 = [&](float* S, int ps, const float (&v)[4]) {
#pragma unroll
        for (int x = 0; x < 4; ++x) S[(sk + ps * 16) * LGF_LDK + sg + x] = v[x];
    };
    auto multiply_chunk = [&]() __attribute__((always_inline)) {
#pragma unroll
        for (int s = 0; s < GK / 4; ++s) {
            float af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = Ah[(4 * s + fk) * LGF_LDK + wr * 32 + i * 16 + fi];
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = Bh[(4 * s + fk) * LGF_LDK + wc * 32 + j * 16 + fi];
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    float pre[2][2][4];
    if (mode == 1) {
        float4 qa[4][2], qb[4][2];
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int gk = c * GK + sk + ps * 16;
                qa[c][ps] = ldq(Cb + (size_t)gk * ld + m0 + sg);
                qb[c][ps] = ldq(Fb + (size_t)gk * ld + n0 + sg);
            }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) pre[i][j][r] = Mi[(size_t)(m0 + wr * 32 + i * 16 + fk * 4 + r) * ld + n0 + wc * 32 + j * 16 + fi];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                float v[4];
                unq(qa[c][ps], v); stage4(Ah, ps, v);
                unq(qb[c][ps], v); stage4(Bh, ps, v);
            }
            __syncthreads();
            multiply_chunk();
            __syncthreads();
        }
    } else {
        // range-checked operand groups (gemm_fetch): a(i, k) = Cb[k][i], b(k, j) = Fb[k][j], k < nloc
        float ra[2][4], rb[2][4];
        auto fetch = [&](int k0) {
#pragma unroll
            for (int ps = 0; ps < 2; ++ps) {
                const int gk = k0 + sk + ps * 16;
                const int ga = m0 + sg, gb = n0 + sg;
                if (a.m.vec && ga + 3 < n && gk < nloc) ld4(Cb + (size_t)gk * ld + ga, ra[ps]);
                else {
#pragma unroll
                    for (int x = 0; x < 4; ++x) ra[ps][x] = (ga + x < n && gk < nloc) ? Cb[(size_t)gk * ld + ga + x] : 0.f;
                }
                if (a.m.vec && gb + 3 < n && gk < nloc) ld4(Fb + (size_t)gk * ld + gb, rb[ps]);
                else {
#pragma unroll
                    for (int x = 0; x < 4; ++x) rb[ps][x] = (gb + x < n && gk < nloc) ? Fb[(size_t)gk * ld + gb + x] : 0.f;
                }
            }
        };
        if (mode == 2) fetch(0);
        for (int c = 0; c < kc; ++c) {
            if (mode == 2) {
#pragma unroll
                for (int ps = 0; ps < 2; ++ps) { stage4(Ah, ps, ra[ps]); stage4(Bh, ps, rb[ps]); }
            }
            __syncthreads();
            if (mode == 2) {
                if (c + 1 < kc) fetch((c + 1) * GK);
                multiply_chunk();
            }
            __syncthreads();
        }
    }

    // ---- epilogue: the expressions of ProbLgUpdate (epi / epi4 / epi4p), tile by tile ----
    if (valid) {
        auto put = [&](float* dst, float v) { if (next_diag && !(ADKF_LGF_EXP & 1)) st_sc1(dst, v); else *dst = v; };
        auto epi = [&](int i, int j, float ac) {
            float* dst = Mi + (size_t)i * ld + j;
            const bool pi = in_p(i), pj = in_p(j);
            float v;
            if (!pi && !pj) v = *dst - ac;
            else if (pi && pj) v = -Dv[(i - p0) * LB + (j - p0)];
            else if (pi) v = Fb[(size_t)(i - p0) * ld + j];
            else v = Fb[(size_t)(j - p0) * ld + i];
            put(dst, v);
            if (ti < tj) put(Mi + (size_t)j * ld + i, v);
        };
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int gi0 = m0 + wr * 32 + i * 16 + fk * 4, gj = n0 + wc * 32 + j * 16 + fi;
                if (mode == 1) {
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { v[r] = pre[i][j][r] - acc[i][j][r]; put(Mi + (size_t)(gi0 + r) * ld + gj, v[r]); }
                    if (ti < tj) {
                        float* mp = Mi + (size_t)gj * ld + gi0;
                        if (next_diag && !(ADKF_LGF_EXP & 1)) { st_sc1(mp, v[0]); st_sc1(mp + 1, v[1]); st_sc1(mp + 2, v[2]); st_sc1(mp + 3, v[3]); }
                        else *reinterpret_cast<float4*>(mp) = make_float4(v[0], v[1], v[2], v[3]);
                    }
                } else if (gi0 + 3 < n && gj < n && a.m.vec && ti != tj && !next_diag) {
                    // four rows of one column, the mirror image as ONE 16-byte store (ProbLgUpdate::epi4)
                    const bool pi = in_p(gi0), pj = in_p(gj);
                    float v[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float* dst = Mi + (size_t)(gi0 + r) * ld + gj;
                        if (!pi && !pj) v[r] = *dst - acc[i][j][r];
                        else if (pi && pj) v[r] = -Dv[(gi0 + r - p0) * LB + (gj - p0)];
                        else if (pi) v[r] = Fb[(size_t)(gi0 + r - p0) * ld + gj];
                        else v[r] = Fb[(size_t)(gj - p0) * ld + gi0 + r];
                        *dst = v[r];
                    }
                    *reinterpret_cast<float4*>(Mi + (size_t)gj * ld + gi0) = make_float4(v[0], v[1], v[2], v[3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (gi0 + r < n && gj < n) epi(gi0 + r, gj, acc[i][j][r]);
                }
            }
    }

    // ---- the next diagonal block: whoever completes it sweeps it ----
    if (nd == 0) return;
    const int mine = pair < nd ? 1 : 0;          // tiles of the next diagonal block in this workgroup (uniform)
    if (mine == 0) return;
#if (ADKF_LGF_EXP & 4)
    if (!direct) return;
    s_last = 1;
#else
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave: its write-through stores have left
    __syncthreads();
    if (tid == 0) {
        const int before = __hip_atomic_fetch_add(a.cnt + t, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (before + mine == nd) ? 1 : 0;
        if (last) {
            if (!(ADKF_LGF_EXP & 2)) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");   // buffer_inv sc1: this CU's L1 drops what it holds of the block
            __hip_atomic_store(a.cnt + t, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        }
        s_last = last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the invalidate has completed before anybody passes the barrier
    }
#endif
    __syncthreads();
    if (!s_last) return;

    const int q0 = p0 + LB, nnext = min(LB, n - q0);
    if (nnext <= 0) return;                      // (a ragged task that ends before the next block)
    float m[RB][CB];
    const float* blk = Mi + (size_t)q0 * ld + q0;
    const bool vec = rows_aligned16(blk, ld);
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int i = SW::row(r);
        load_segment<CB>(blk + (size_t)i * ld, SW::col(0), nnext, i < nnext, vec, m[r]);
#pragma unroll
        for (int c = 0; c < CB; ++c)
            if (i == SW::col(c) && i >= nnext) m[r][c] = 1.f;
    }
    __syncthreads();
    if (a.prio > 0) __builtin_amdgcn_s_setprio(3);   // the chain's instructions go first on SIMDs it shares with an updating workgroup
    SW::run(m, nnext, sm);
    if (a.prio > 0) __builtin_amdgcn_s_setprio(0);
    float logdet;
    const int info = SW::finish(nnext, sm, logdet);
    float* Dn = a.Dinv_next + (size_t)t * LB * LB;
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        float neg[CB];
#pragma unroll
        for (int c = 0; c < CB; ++c) neg[c] = -m[r][c];
        store_segment<CB>(Dn + SW::row(r) * LB, SW::col(0), LB, true, true, neg);
    }
    float plo = INFINITY, phi = 0.f;
    if (tid < nnext) { plo = phi = sm.pivs[tid]; }
    plo = wave_min(plo); phi = wave_max(phi);
    __syncthreads();
    if ((tid & 63) == 0) { sm.red[2 * (tid >> 6)] = plo; sm.red[2 * (tid >> 6) + 1] = phi; }
    __syncthreads();
    if (tid == 0) {
        a.m.logdet[t] = a.m.logdet[t] + logdet;            // (block step + 1 >= 1: the running values of k_lg_diag at step 0 exist)
        const int prev = a.m.info[t];
        a.m.info[t] = prev != 0 ? prev : (info != 0 ? q0 + info : 0);
        plo = fminf(sm.red[0], sm.red[2]); phi = fmaxf(sm.red[1], sm.red[3]);
        if (a.m.pext) {
            a.m.pext[2 * t] = fminf(a.m.pext[2 * t], plo);
            a.m.pext[2 * t + 1] = fmaxf(a.m.pext[2 * t + 1], phi);
        }
    }
}

}  // namespace adkf
