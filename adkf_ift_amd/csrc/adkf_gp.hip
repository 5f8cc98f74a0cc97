// libadkf_gp.so - host side of the C ABI declared in include/adkf_gp.h: argument checks, workspace
// carving and the kernel pipeline of each entry point.  gfx950 only; no allocation, no synchronisation
// (except adkf_check_info), everything enqueued on the caller's stream.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>

#include "../../include/adkf_gp.h"
#include "ard.h"
#include "pna.h"
#include "readout.h"
#include "block.h"
#include "outer_step.h"
#include "refine64.h"
#include "hyper.h"
#include "large_fused.h"
#include "gemm_x3.h"
#include "dense_x3.h"
#if ADKF_VARIANT_DZ   // A/B experiment only (measured slower than the two ProbDZ launches: see its header)
#include "../../tools/variants/dz.h"
#endif

using namespace adkf;

#if ADKF_EVAL_STAMP
extern "C" __device__ unsigned long long adkf_eval_stamps[16] = {};
extern "C" int adkf_read_eval_stamps(unsigned long long* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(adkf_eval_stamps), sizeof(unsigned long long) * 16); }
#endif

namespace {

constexpr int MAX_POINTS = 4096;  // <= 128: register-resident sweep (inner.h); above: blocked sweep through L2/HBM (large.h)
constexpr int REG_POINTS = 128;

inline size_t align_up(size_t x) { return (x + 255) & ~size_t(255); }

inline int grid_for(int T, int tiles) { return ((T + 7) / 8) * 8 * tiles; }
inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Output-tile edge of the batched GEMMs.  The 128 x 128 variant (gemm.h) halves the operand traffic but leaves one
// wave per SIMD: measured slower at every stage of C2 (ProbDist 33 -> 49 us, ProbP 22 -> 32 us, profiles/ notes in
// DESIGN.md), so 64 x 64 (four co-resident workgroups per CU) is used throughout.
inline int tile_edge(int, int) { return GT; }
inline int tiles_of(int M, int N) { const int e = tile_edge(M, N); return ceil_div(M, e) * ceil_div(N, e); }

// Which problems run on the BF16 matrix pipe (gemm_x3.h: FP32 products out of three-way split operands).  ADKF_X3=0 (read once) sends
// them back to the FP32-input MFMA kernel for A/B runs.
template <class P> struct use_x3 : std::false_type {};
// (ProbDZ ran on it as well - every parity test green, MN-contiguous staging with a column per lane - at the same time as on the FP32
// pipe, 62.5 + 54.4 against 61 + 56.5 us at C2: it is bound by its strided W_qs^T operand and the staging, not by the matrix pipe; left
// on the FP32 form.)
template <> struct use_x3<ProbDistMulti> : std::true_type {};
// (The N^3 products of the multi-launch outer stage beyond 128 points - ProbP, ProbC, ProbS, ProbOC, ProbMA, ProbMixed - were tried on
// it as well: the same 1.99 ms for the eleven products of a C5 step, profiles/r05_c5_x3_kernel_stats.csv - at 64 x 64 tiles and
// K = 1024 they wait for their operands, 6.9 TB/s out of L2 / MALL, not for the matrix pipe.)
bool x3_enabled() {
    static const bool v = [] { const char* e = getenv("ADKF_X3"); return !(e && e[0] == '0'); }();
    return v;
}

// Feature dimensions below one K chunk stay on the FP32 form: nothing to gain there (the chunk is mostly padding), and the stress
// suite's low-dimensional clustered tasks (d = 2, 3: cond 2e2 .. 6e2, where the float32 restatement of the reference itself is
// 1e-4 .. 3e-4 from float64) keep the arithmetic their tolerances were measured with.
inline bool x3_for(int d) { return x3_enabled() && d >= GK; }

template <class P>
void launch_gemm(const P& p, int T, int M, int N, hipStream_t st, bool x3 = false) {
    const int tm = ceil_div(M, GT), tn = ceil_div(N, GT);   // (k_bgemm<P, GTL> is not instantiated: see tile_edge)
    if constexpr (use_x3<P>::value) {
        if (x3 && x3_enabled()) { k_bgemm3<P, GT, 256><<<((T + 7) / 8) * 8 * tm * tn, 256, 0, st>>>(p, T, tm, tn); return; }
    }
    k_bgemm<P, GT><<<((T + 7) / 8) * 8 * tm * tn, 256, 0, st>>>(p, T, tm, tn);
}

struct Workspace {
    float *mean, *D2ss, *D2qs, *D2qq, *Ainv, *P, *C, *S, *OC, *Wss, *Wqs, *Wqq, *vecs, *scal, *part_oc, *part_ma, *l0;
    // blocked path only (max(ns, nq) > REG_POINTS)
    float *lg_Dinv, *lg_C, *lg_F, *lg_logdet, *lg_part, *lg_pext;   // lg_Dinv: [2, T, LB, LB] (the fused block step alternates between the two)
    int32_t *lg_info, *lg_med, *lg_cnt;  // lg_med: prefix[T], rank[T], hist[T, 256]; lg_cnt: [T] arrival counters of large_fused.h
    FitShared* lg_fit;
    double* w64; size_t w64_stride;   // float64 region of the ill-conditioned-task path (refine64.h); null beyond R64_MAXN points
    int vld, nt_oc, nt_ma;
    int lg_mode;       // ADKF_BATCH_LG_UNFUSED / ADKF_BATCH_LG_FUSED of the batch this view was carved for: -1 three launches, +1 fused, 0 by size
    size_t bytes;
};

Workspace carve(void* base, int T, int ns, int nq, int d) {
    Workspace w;
    w.lg_mode = 0;
    size_t off = 0;
    auto take = [&](size_t nfloat) { float* p = base ? reinterpret_cast<float*>(static_cast<char*>(base) + off) : nullptr; off += align_up(nfloat * sizeof(float)); return p; };
    const size_t Tz = (size_t)T;
    w.vld = ns > nq ? ns : nq;
    w.nt_oc = nq > 0 ? tiles_of(nq, ns) : 0;   // per-tile partial reductions of ProbOC / ProbMA
    w.nt_ma = tiles_of(ns, ns);
    w.mean = take(Tz * d);
    w.D2ss = take(Tz * ns * ns);
    w.D2qs = take(Tz * nq * ns);
    w.D2qq = take(Tz * nq * nq);
    w.Ainv = take(Tz * ns * ns);
    w.P = take(Tz * ns * ns);
    w.C = take(Tz * nq * ns);
    w.S = take(Tz * nq * nq);
    w.OC = take(Tz * nq * ns);
    w.Wss = take(Tz * ns * ns);
    w.Wqs = take(Tz * nq * ns);
    w.Wqq = take(Tz * nq * nq);
    w.vecs = take(Tz * NVEC * w.vld);
    w.scal = take(Tz * NSCAL);
    w.part_oc = take(Tz * (w.nt_oc > 0 ? w.nt_oc : 1) * 4);
    w.part_ma = take(Tz * w.nt_ma * 4);
    w.l0 = take(Tz);
    w.lg_Dinv = w.lg_C = w.lg_F = w.lg_logdet = w.lg_part = w.lg_pext = nullptr; w.lg_info = w.lg_med = w.lg_cnt = nullptr; w.lg_fit = nullptr;
    if (w.vld > REG_POINTS) {
        w.lg_Dinv = take(2 * Tz * LB * LB);
        w.lg_C = take(Tz * LB * w.vld);
        w.lg_F = take(Tz * LB * w.vld);
        w.lg_logdet = take(Tz);
        w.lg_pext = take(Tz * 2);
        const size_t tq = (size_t)((nq + GT - 1) / GT) * ((nq + GT - 1) / GT);
        const size_t ts = (size_t)((ns + GT - 1) / GT) * ((ns + GT - 1) / GT);
        w.lg_part = take(Tz * (ts > tq ? ts : tq) * 8);
        w.lg_info = reinterpret_cast<int32_t*>(take(Tz));
        w.lg_cnt = reinterpret_cast<int32_t*>(take(Tz));
        w.lg_med = reinterpret_cast<int32_t*>(take(Tz * 258));
        w.lg_fit = reinterpret_cast<FitShared*>(take(Tz * ((sizeof(FitShared) + 3) / 4)));
    }
    w.w64 = nullptr; w.w64_stride = 0;
    // the float64 region of refine64.h is carved for EVERY task (any of them may turn out ill-conditioned): 8 refine64_doubles(ns, nq)
    // bytes per task - 1.3 MB at 128 points, 5.2 MB at 256, 82 MB at 1024 (1.7 x the float32 part of the workspace).  ADKF_R64_MAXN
    // (read once) lowers the largest batch that gets one, e.g. 256: tasks beyond it stay on the float32 path whatever their conditioning
    static const int r64_maxn = [] { const char* e = getenv("ADKF_R64_MAXN"); const int v = e ? atoi(e) : R64_MAXN; return v < R64_MAXN ? v : R64_MAXN; }();
    if (w.vld <= r64_maxn) {
        w.w64_stride = refine64_doubles(ns, nq);
        w.w64 = reinterpret_cast<double*>(take(2 * Tz * w.w64_stride));
    }
    w.bytes = off;
    return w;
}

Workspace carve_for(const adkf_batch_t* b, void* ws) {
    Workspace w = carve(ws, b->T, b->ns_max, b->nq_max, b->d);
    w.lg_mode = (b->flags & ADKF_BATCH_LG_UNFUSED) ? -1 : (b->flags & ADKF_BATCH_LG_FUSED) ? 1 : 0;
    return w;
}

int check_batch(const adkf_batch_t* b, bool need_query) {
    (void)hipGetLastError();  // a stale error left by another library on this thread must not be blamed on our launches
    if (!b || b->T <= 0 || b->ns_max <= 0 || b->d <= 0 || !b->Z_s) return ADKF_E_BADARG;
    if (b->kernel != ADKF_KERNEL_RBF && b->kernel != ADKF_KERNEL_MATERN52) return ADKF_E_BADARG;
    if (b->ns_max > MAX_POINTS) return ADKF_E_SIZE;
    if (need_query) {
        if (b->nq_max <= 0 || !b->Z_q) return ADKF_E_BADARG;
        if (b->nq_max > MAX_POINTS) return ADKF_E_SIZE;
    }
    return 0;
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// 16-byte vector loads are legal when every leading dimension is a multiple of 4 floats and every base is aligned
// (the workspace carve keeps 256-byte alignment; per-task offsets are then multiples of 16 bytes too).
bool vec_ok(const adkf_batch_t* b, const Workspace& w) {
    if ((b->ns_max & 3) || (b->nq_max & 3) || (b->d & 3) || (w.vld & 3)) return false;
    return aligned16(b->Z_s) && aligned16(b->Z_q) && aligned16(w.mean);
}

TaskView make_tv(const adkf_batch_t* b, const Workspace& w, bool with_query) {
    TaskView tv;
    tv.n_s = b->n_s; tv.n_q = with_query ? b->n_q : nullptr;
    tv.ns_ld = b->ns_max; tv.nq_ld = with_query ? b->nq_max : 0; tv.vld = w.vld; tv.kind = b->kernel;
    tv.scal = w.scal; tv.vecs = w.vecs;
    tv.vec = vec_ok(b, w);
    return tv;
}

thread_local hipError_t g_last_hip_error = hipSuccess;  // diagnostics only: what ADKF_E_LAUNCH was about (adkf_last_hip_error)
#define LAUNCH_OK() do { const hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) { g_last_hip_error = e_; return ADKF_E_LAUNCH; } } while (0)

inline bool has_query(const adkf_batch_t* b) { return b->nq_max > 0 && b->Z_q != nullptr; }

// Stage A: centring, squared distances.  Skipped when the caller promises (ADKF_BATCH_REUSE_DIST) that
// this workspace already holds them for exactly this batch.
// parts: 1 = the support block (mean, norms, D2ss), 2 = the query blocks (needs the support mean / norms in place),
// 4 = the features are already centred and w.mean holds zeros (ARD: Z~ = (Z - mu) / l has zero column mean by construction).
// (The squared row norms are summed inside the distance GEMM while it stages its operands: no separate pass.)
int stage_dist(const adkf_batch_t* b, const Workspace& w, bool with_query, hipStream_t st, int parts = 3) {
    if (b->flags & ADKF_BATCH_REUSE_DIST) return 0;
    const int T = b->T, ns = b->ns_max, nq = with_query ? b->nq_max : 0, d = b->d;
    if (!(parts & 2)) with_query = false;
    if ((parts & 1) && !(parts & 4)) k_colmean<<<dim3(ceil_div(d, 64), T), 256, 0, st>>>(b->Z_s, b->n_s, ns, d, w.mean, T);
    ProbDist p;
    p.mean = w.mean; p.d = d; p.vec = vec_ok(b, w);
    ProbDistMulti pm;
    pm.vec = p.vec; pm.end0 = pm.end1 = 0; pm.tn0 = pm.tn1 = pm.tn2 = 1;
    int nblk = 0, total = 0;
    auto add = [&](const ProbDist& q, int M, int N) {
        const int tn = ceil_div(N, GT), tiles = ceil_div(M, GT) * tn;
        if (nblk == 0) { pm.s0 = q; pm.tn0 = tn; pm.end0 = pm.end1 = total + tiles; pm.s1 = pm.s2 = q; }
        else if (nblk == 1) { pm.s1 = q; pm.tn1 = tn; pm.end1 = total + tiles; pm.s2 = q; }
        else { pm.s2 = q; pm.tn2 = tn; }
        total += tiles; ++nblk;
    };
    if (parts & 1) {
        p.X = b->Z_s; p.Y = b->Z_s; p.n_x = b->n_s; p.n_y = b->n_s; p.x_ld = ns; p.y_ld = ns; p.symmetric = true; p.D2 = w.D2ss;
        add(p, ns, ns);
    }
    if (with_query) {
        p.X = b->Z_q; p.Y = b->Z_s; p.n_x = b->n_q; p.n_y = b->n_s; p.x_ld = nq; p.y_ld = ns; p.symmetric = false; p.D2 = w.D2qs;
        add(p, nq, ns);
        p.X = b->Z_q; p.Y = b->Z_q; p.n_x = b->n_q; p.n_y = b->n_q; p.x_ld = nq; p.y_ld = nq; p.symmetric = true; p.D2 = w.D2qq;
        add(p, nq, nq);
    }
    if (nblk > 0) {
        if (x3_for(d)) k_bgemm3<ProbDistMulti, GT, 256><<<grid_for(T, total), 256, 0, st>>>(pm, T, 1, total);
        else k_bgemm<ProbDistMulti, GT><<<grid_for(T, total), 256, 0, st>>>(pm, T, 1, total);
    }
    LAUNCH_OK();
    return 0;
}

int num_cus() {
    static const int n = [] {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
        return cus;
    }();
    return n;
}

template <int NMAX, int NT, bool LOW>
void launch_inner_kl(const InnerArgs& a, hipStream_t st) {
    constexpr size_t cache_bytes = sizeof(float) * NMAX * NMAX;   // the kappa' u cache of inner.h (one float per matrix element), or D^2 (LOW)
    static const bool attr_set = [] {   // 64 KB of dynamic LDS on top of the static part needs the opt-in
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_inner<NMAX, NT, 0, LOW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cache_bytes);
        hipFuncSetAttribute(reinterpret_cast<const void*>(&k_inner<NMAX, NT, 1, LOW>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cache_bytes);
        return true;
    }();
    (void)attr_set;
    if (a.kind == ADKF_KERNEL_RBF) k_inner<NMAX, NT, 0, LOW><<<grid_for(a.T, 1), NT, cache_bytes, st>>>(a);
    else k_inner<NMAX, NT, 1, LOW><<<grid_for(a.T, 1), NT, cache_bytes, st>>>(a);
}

// ADKF_INNER_LOWREG (read once): 1 / 0 force the two-tasks-per-CU variant of the 128-point fit on / off; unset: taken when the
// batch has more tasks than the chip has CUs (up to that every task has a CU to itself and the resident variant is 20 % faster;
// beyond it the resident variant needs a second round of workgroups, the low-register one runs two tasks per CU side by side)
template <int NMAX, int NT>
void launch_inner_k(const InnerArgs& a, hipStream_t st) {
    if constexpr (NMAX == 128) {
        static const int forced = [] { const char* e = getenv("ADKF_INNER_LOWREG"); return e ? atoi(e) : -1; }();
        const bool low = forced >= 0 ? forced != 0 : a.T > num_cus();
        if (low) { launch_inner_kl<NMAX, NT, true>(a, st); return; }
    }
    launch_inner_kl<NMAX, NT, false>(a, st);
}

// ADKF_LG_FUSED (read once): 0 keeps the three launches per block step (k_lg_diag, panel, update) for A/B runs; default: the update
// of block step k and the sweep of block step k + 1 share a launch (large_fused.h)
bool lg_fused() {
    static const bool on = [] { const char* e = getenv("ADKF_LG_FUSED"); return !e || atoi(e) != 0; }();
    return on;
}

LgMat lg_mat(const Workspace& w, float* M, int ld, const int32_t* n_arr, const FitShared* fit, int T) {
    LgMat m;
    m.M = M; m.ld = ld; m.n_arr = n_arr; m.fit = fit;
    m.Dinv = w.lg_Dinv; m.Cbuf = w.lg_C; m.Fbuf = w.lg_F; m.logdet = w.lg_logdet; m.pext = w.lg_pext; m.info = w.lg_info;
    // by size: from four block steps on (tools/lgf_bench.hip, profiles/r05_lgf_bench.txt: 8 x 1024 points 0.85 x the time of the three
    // launches, 16 x 1024 0.90 x; but 64 x 256, 5 x 515 and 3 x 300 points 1.06 - 1.10 x: with two or three block steps the sweep that
    // rides in the update launch is most of that launch)
    const bool fused = lg_fused() && (w.lg_mode > 0 || (w.lg_mode == 0 && ld >= 4 * LB));
    m.cnt = fused ? w.lg_cnt : nullptr;
    m.T = T; m.vec = (ld & 3) == 0;
    return m;
}

// M -> -(M^-1) in place by 128-pivot block steps (large.h; large_fused.h)
void lg_sweep(const LgMat& m0, hipStream_t st) {
    const int nb = ceil_div(m0.ld, LB), tn = ceil_div(m0.ld, GT);
    LgMat m = m0;
    if (m.cnt) {
        // D(0) | P(0) | U(0) + D(1) | P(1) | U(1) + D(2) | ... | P(nb - 1) | U(nb - 1): 2 nb + 1 launches instead of 3 nb
        float* dinv[2] = {m0.Dinv, m0.Dinv + (size_t)m0.T * LB * LB};
        const int npair = lgf_npair(tn);
        k_lg_diag<<<grid_for(m.T, 1), 512, 0, st>>>(m, 0);
        for (int step = 0; step < nb; ++step) {
            m.Dinv = dinv[step & 1];
            ProbLgPanel pp; pp.m = m; pp.step = step;
            k_bgemm<ProbLgPanel><<<grid_for(m.T, 2 * tn), 256, 0, st>>>(pp, m.T, 2, tn);
            LgStepArgs sa{m, dinv[(step + 1) & 1], m.cnt, step, tn, npair, step + 1 < nb ? 1 : 0, LGF_STAGGER, LGF_PRIO};
            k_lg_update_sweep<<<grid_for(m.T, npair), LGF_NT, 0, st>>>(sa);
        }
        return;
    }
    for (int step = 0; step < nb; ++step) {
        k_lg_diag<<<grid_for(m.T, 1), 512, 0, st>>>(m, step);
        ProbLgPanel pp; pp.m = m; pp.step = step;
        k_bgemm<ProbLgPanel><<<grid_for(m.T, 2 * tn), 256, 0, st>>>(pp, m.T, 2, tn);
        ProbLgUpdate pu; pu.m = m; pu.step = step; pu.tri = tn * (tn + 1) / 2;   // workgroups for the tiles on or above the diagonal only
        k_bgemm<ProbLgUpdate><<<grid_for(m.T, pu.tri), 256, 0, st>>>(pu, m.T, tn, tn);
    }
}

// Convergence-mode early exit for the fits that are a sequence of launches (blocked path, ARD): every POLL_EVERY
// evaluations the number of unfinished tasks goes to pinned host memory and the stream is synchronised, so a fit that
// converges after 15 evaluations does not enqueue the other 185 rounds of (skipped) kernels.  Never in exact-evals mode
// (deterministic work, no synchronisation) and never while the stream is being captured into a graph.
constexpr int POLL_EVERY = 8;

__global__ void k_count_unfinished(const char* base, size_t stride, size_t phase_offset, int T, int done_value, int32_t* out) {
    int c = 0;
    for (int t = threadIdx.x; t < T; t += blockDim.x) c += *reinterpret_cast<const int*>(base + (size_t)t * stride + phase_offset) != done_value;
    c = wave_sum_i(c);
    if (threadIdx.x == 0) *out = c;
}

struct FitPoll {
    bool enabled = false;
    int32_t* host = nullptr;
    int32_t* dev;
    FitPoll(bool convergence_mode, int max_evals, int32_t* dev_counter, hipStream_t st) : dev(dev_counter) {
        if (!convergence_mode || max_evals <= 2 * POLL_EVERY) return;
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(st, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) return;
        thread_local int32_t* pinned = nullptr;
        if (!pinned && hipHostMalloc(reinterpret_cast<void**>(&pinned), sizeof(int32_t), hipHostMallocDefault) != hipSuccess) return;
        host = pinned;
        enabled = true;
    }
    // true when every task has finished (call after the advance kernel of evaluation e).  every_after: once POLL_EVERY rounds are
    // through, poll every that many (the CG loop: a round of empty launches costs more than a poll once most tasks have converged)
    bool finished(int e, const void* state, size_t stride, size_t phase_offset, int T, hipStream_t st, int done_value = PH_DONE, int every_after = POLL_EVERY) {
        if (!enabled) return false;
        if (e + 1 < POLL_EVERY || (e + 1 - POLL_EVERY) % every_after != 0) return false;
        k_count_unfinished<<<1, 64, 0, st>>>(static_cast<const char*>(state), stride, phase_offset, T, done_value, dev);
        if (hipMemcpyAsync(host, dev, sizeof(int32_t), hipMemcpyDeviceToHost, st) != hipSuccess) return false;
        if (hipStreamSynchronize(st) != hipSuccess) return false;
        return *host == 0;
    }
};

int launch_inner_large(const InnerArgs& a, const Workspace& w, hipStream_t st) {
    LgInner li;
    li.in = a; li.fit = w.lg_fit; li.part = w.lg_part;
    li.tiles_1d = ceil_div(a.ld, GT); li.ntiles = li.tiles_1d * li.tiles_1d;
    li.mat = lg_mat(w, a.Ainv, a.ld, a.n_s, w.lg_fit, a.T);
    LgMatvecArgs mv{li.mat, a.y_s, (size_t)a.ld, a.vecs + (size_t)V_ALPHA * a.vld, (size_t)NVEC * a.vld, -1.f};
    k_lg_begin<<<ceil_div(a.T, 64), 64, 0, st>>>(li);
    const int n_evals = a.max_evals > 0 ? a.max_evals : 1;
    FitPoll poll(a.max_evals > 0 && !a.exact_evals, a.max_evals, w.lg_info, st);   // lg_info[0] is free between block sweeps
    for (int e = 0; e < n_evals; ++e) {
        k_lg_build<<<grid_for(a.T, li.ntiles), 256, 0, st>>>(li);
        lg_sweep(li.mat, st);
        k_lg_matvec<<<dim3(ceil_div(a.ld, 4), a.T), 256, 0, st>>>(mv);
        k_lg_traces<<<grid_for(a.T, li.ntiles), 256, 0, st>>>(li);
        k_lg_advance<<<a.T, 256, 0, st>>>(li);
        if (poll.finished(e, w.lg_fit, sizeof(FitShared), offsetof(FitShared, phase), a.T, st)) break;
    }
    LAUNCH_OK();
    return 0;
}

// Stage B (and the fit): dispatch on the padded support size.
int launch_inner(InnerArgs a, const Workspace& w, hipStream_t st) {
    if (a.ld > REG_POINTS) return launch_inner_large(a, w, st);
    if (a.ld <= 16) launch_inner_k<16, 256>(a, st);
    else if (a.ld <= 32) launch_inner_k<32, 256>(a, st);
    else if (a.ld <= 64) launch_inner_k<64, 256>(a, st);
    else launch_inner_k<128, 512>(a, st);
    LAUNCH_OK();
    return 0;
}

// ADKF_R64_THRESHOLD (read once) moves the switch-over for experiments: 0 sends every task through float64, a huge value none
float r64_threshold() {
    static const float thresh = [] {
        const char* e = getenv("ADKF_R64_THRESHOLD");
        return e ? (float)atof(e) : R64_THRESHOLD;
    }();
    return thresh;
}

// Ill-conditioned tasks redo the factorisation-type stages in float64 (refine64.h); everybody else leaves the kernel after
// reading two scalars.  level: 0 = inner quantities (A^-1, alpha, scalars), 1 = + C, mu (prediction), 2 = + S^-1, e, f_out.
// (s + noise) max_i (A^-1)_ii above which C and alpha get one step of float32 iterative refinement (ADKF_REFINE32_THRESHOLD moves it)
float refine32_threshold() {
    static const float thresh = [] {
        const char* e = getenv("ADKF_REFINE32_THRESHOLD");
        return e ? (float)atof(e) : 3.f;
    }();
    return thresh;
}

void launch_alpha_refine(const TaskView& tv, const adkf_batch_t* b, const Workspace& w, hipStream_t st) {
    AlphaRefineArgs aa{tv, w.Ainv, w.D2ss, b->y_s, w.vecs, refine32_threshold(), b->T};
    k_alpha_refine<<<grid_for(b->T, 1), SMALL_NT, 0, st>>>(aa);
}

// C = K_qs A^-1 followed, for the tasks that need it, by one refinement step (R lives in w.OC, which ProbOC fills later)
void launch_c(const TaskView& tv, const adkf_batch_t* b, const Workspace& w, hipStream_t st) {
    const int T = b->T, ns = b->ns_max, nq = b->nq_max;
    ProbC pc; pc.tv = tv; pc.Ainv = w.Ainv; pc.D2qs = w.D2qs; pc.C = w.C;
    launch_gemm(pc, T, nq, ns, st);
    ProbCres pr; pr.tv = tv; pr.C = w.C; pr.D2ss = w.D2ss; pr.D2qs = w.D2qs; pr.R = w.OC; pr.thresh = refine32_threshold();
    launch_gemm(pr, T, nq, ns, st);
    ProbCfix pf; pf.tv = tv; pf.R = w.OC; pf.Ainv = w.Ainv; pf.C = w.C; pf.thresh = refine32_threshold();
    launch_gemm(pf, T, nq, ns, st);
}

bool refine64_lds_optin() {
    // up to R64_LDS_POINTS points the float64 inverses run in LDS: 128 KB of dynamic shared memory (one workgroup per CU then; the
    // kernels are a two-scalar test for everybody but the flagged tasks).  Without the opt-in the inverses work in global memory.
    static const bool ok = [] {
        const int bytes = (int)(sizeof(double) * R64_LDS_POINTS * R64_LDS_POINTS);
        return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_refine64), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess &&
               hipFuncSetAttribute(reinterpret_cast<const void*>(&k_tail64), hipFuncAttributeMaxDynamicSharedMemorySize, bytes) == hipSuccess;
    }();
    if (!ok) (void)hipGetLastError();
    return ok;
}

// ADKF_R64_STOP (read once; diagnostics, tools/history/r64_phases.sh): the float64 path leaves after that phase - results are then garbage
int r64_stop() {
    static const int stop = [] {
        const char* e = getenv("ADKF_R64_STOP");
        const int v = e ? atoi(e) : 0;
        if (v != 0) fprintf(stderr, "libadkf_gp: ADKF_R64_STOP=%d is set - the float64 path leaves after that phase and its RESULTS ARE GARBAGE (phase-timing diagnostics only)\n", v);
        return v;
    }();
    return stop;
}

Refine64Args refine_args(const TaskView& tv, const adkf_batch_t* b, const Workspace& w, bool with_hessian, int level, float* f_out,
                         int32_t* info, float* f_in, float* g_in, float* gnorm, size_t& lds_bytes) {
    const bool lds_inv = refine64_lds_optin();   // (beyond R64_LDS_POINTS the diagonal blocks of the blocked inverse live there)
    // (always the full R64_LDS_POINTS^2 doubles - 128 KB: the staged products of refine64.h work in blocks of that edge whatever the
    // batch size; one workgroup per CU, which is what this path runs at anyway)
    lds_bytes = lds_inv ? sizeof(double) * (size_t)R64_LDS_POINTS * R64_LDS_POINTS : 0;
    return Refine64Args{tv, b->Z_s, b->Z_q, b->d, b->y_s, b->y_q, b->priors, w.Ainv, with_hessian ? w.P : nullptr, level >= 1 ? w.C : nullptr,
                        level >= 2 ? w.S : nullptr, w.vecs, w.scal, f_out, info, f_in, g_in, gnorm, w.w64, w.w64_stride, r64_threshold(), b->T,
                        with_hessian ? 1 : 0, level, lds_inv ? 1 : 0, r64_stop()};
}

// Ill-conditioned tasks redo the factorisation-type stages in float64 (refine64.h); everybody else leaves the kernel after
// reading two scalars.  level: 0 = inner quantities (A^-1, alpha, scalars), 1 = + C, mu (prediction).  (Level 2 - + S^-1, e, f_out -
// runs inside k_tail64 at the end of the hypergradient pipeline.)
void launch_refine(const TaskView& tv, const adkf_batch_t* b, const Workspace& w, bool with_hessian, int level, float* f_out,
                   int32_t* info, hipStream_t st, float* f_in = nullptr, float* g_in = nullptr, float* gnorm = nullptr) {
    if (!w.w64) return;
    size_t lds_bytes;
    const Refine64Args ra = refine_args(tv, b, w, with_hessian, level, f_out, info, f_in, g_in, gnorm, lds_bytes);
    k_refine64<<<b->T, R64_NT, lds_bytes, st>>>(ra);
}

// 64 < max(support, query) <= 128: the outer / hypergradient stage of a task runs as ONE workgroup (hyper.h).  ADKF_FUSED_OUTER=0
// (read once) keeps the sixteen-launch pipeline for A/B measurements.
bool use_fused_outer(int ns, int nq) {
    static const bool enabled = [] { const char* e = getenv("ADKF_FUSED_OUTER"); return !e || atoi(e) != 0; }();
    // ADKF_FUSED_OUTER_MIN (read once, experiments): smallest max(support, query) that takes the one-kernel stage; default 1 - round 5
    // sends the small shapes (C1, 16 / 32 / 64-shot tasks) through the ragged instance too: one launch instead of sixteen
    static const int min_pts = [] { const char* e = getenv("ADKF_FUSED_OUTER_MIN"); return e ? atoi(e) : 1; }();
    static const bool optin = [] {
        bool ok = true;
        for (const void* f : {reinterpret_cast<const void*>(&k_hyper<true, 0>), reinterpret_cast<const void*>(&k_hyper<true, 1>),
                              reinterpret_cast<const void*>(&k_hyper<false, 0>), reinterpret_cast<const void*>(&k_hyper<false, 1>)})
            ok = ok && hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)HY_LDS_BYTES) == hipSuccess;
        return ok;
    }();
    if (!optin) (void)hipGetLastError();
    const int hi = ns > nq ? ns : nq;
    return enabled && optin && hi >= min_pts && hi <= HY_N;
}

int launch_outer_factor(const OuterArgs& a, const Workspace& w, int nq, hipStream_t st) {
    if (nq > REG_POINTS) {
        k_lg_resid<<<dim3(ceil_div(nq, 4), a.T), 256, 0, st>>>(a);
        LgMat m = lg_mat(w, a.S, a.tv.nq_ld, a.tv.n_q, nullptr, a.T);
        lg_sweep(m, st);
        LgMatvecArgs mv{m, a.vecs + (size_t)V_R * a.tv.vld, (size_t)NVEC * a.tv.vld, a.vecs + (size_t)V_E * a.tv.vld, (size_t)NVEC * a.tv.vld, -1.f};
        k_lg_matvec<<<dim3(ceil_div(nq, 4), a.T), 256, 0, st>>>(mv);
        const int tn = ceil_div(nq, GT);
        k_lg_negate<<<grid_for(a.T, tn * tn), 256, 0, st>>>(m, tn);
        LgColsumArgs cs{a.C, a.tv.ns_ld, (size_t)a.tv.nq_ld * a.tv.ns_ld, a.tv.n_q, a.tv.nq_ld, a.tv.n_s, a.tv.ns_ld,
                        a.vecs + (size_t)V_E * a.tv.vld, (size_t)NVEC * a.tv.vld, a.vecs + (size_t)V_CTE * a.tv.vld, (size_t)NVEC * a.tv.vld};
        k_lg_colsum<<<dim3(ceil_div(a.tv.ns_ld, 64), a.T), 1024, 0, st>>>(cs);
        LgOuterFin fin{a, w.lg_logdet, w.lg_info, w.lg_pext};
        k_lg_outer_fin<<<a.T, 64, 0, st>>>(fin);
        LAUNCH_OK();
        return 0;
    }
    if (nq <= 16) k_outer_factor<16, 256><<<grid_for(a.T, 1), 256, 0, st>>>(a);
    else if (nq <= 32) k_outer_factor<32, 256><<<grid_for(a.T, 1), 256, 0, st>>>(a);
    else if (nq <= 64) k_outer_factor<64, 256><<<grid_for(a.T, 1), 256, 0, st>>>(a);
    else k_outer_factor<128, 512><<<grid_for(a.T, 1), 512, 0, st>>>(a);
    LAUNCH_OK();
    return 0;
}


InnerArgs inner_args(const adkf_batch_t* b, const Workspace& w, float* phi, int32_t* info) {
    InnerArgs a{};
    a.D2ss = w.D2ss; a.y_s = b->y_s; a.n_s = b->n_s; a.phi = phi; a.priors = b->priors;
    a.Ainv = w.Ainv; a.vecs = w.vecs; a.scal = w.scal; a.info = info;
    a.T = b->T; a.ld = b->ns_max; a.vld = w.vld; a.kind = b->kernel;
    a.max_evals = 0; a.exact_evals = 0; a.gtol = 0.f; a.ftol = 0.f;
    return a;
}

// Stage D..G shared by adkf_outer_nll_value_grad (with_hessian = false) and adkf_ift_hypergrad.
int outer_pipeline(const adkf_batch_t* b, const Workspace& w, const float* phi, int flags, bool with_hessian, float* f_out,
                   float* dZ_s, float* dZ_q, float* g_phi_out, float* v_out, float* H_out, int32_t* info, hipStream_t st) {
    const int T = b->T, ns = b->ns_max, nq = b->nq_max, d = b->d;
    int rc = stage_dist(b, w, true, st);
    if (rc) return rc;
    const bool reuse_inner = (b->flags & ADKF_BATCH_REUSE_INNER) != 0;
    if (!reuse_inner) {
        InnerArgs ia = inner_args(b, w, const_cast<float*>(phi), info);
        rc = launch_inner(ia, w, st);
        if (rc) return rc;
    }
    TaskView tv = make_tv(b, w, true);
    const int tms = ceil_div(ns, GT), tmq = ceil_div(nq, GT);
    const float dirscale = (flags & ADKF_IGNORE_DIRECT_GRAD) ? 0.f : 1.f;
    const float corrscale = (with_hessian && !(flags & ADKF_IGNORE_GRAD_CORRECTION)) ? 1.f : 0.f;
    if (use_fused_outer(ns, nq)) {
        // (reused inner stage: A^-1, alpha and the scalars of phi are in the workspace, info[] is written by this kernel)
        HyperArgs ha{tv, w.Ainv, w.D2ss, w.D2qs, w.D2qq, b->y_s, b->y_q, b->priors, w.Wss, w.Wqs, w.Wqq, w.P, w.OC, w.S, w.vecs, w.scal, f_out, info,
                     g_phi_out, v_out, H_out, T, reuse_inner ? 1 : 0, with_hessian ? 1 : 0, flags, dirscale, corrscale, refine32_threshold()};
        // FULL: every task has exactly 128 support and 128 query points in 16-byte aligned rows (affine addresses, no clamps)
        const bool full = ns == HY_N && nq == HY_N && !b->n_s && !b->n_q && tv.vec;
        const bool rbf = b->kernel == ADKF_KERNEL_RBF;
        if (full && rbf) k_hyper<true, 0><<<grid_for(T, 1), HY_NT, HY_LDS_BYTES, st>>>(ha);
        else if (full) k_hyper<true, 1><<<grid_for(T, 1), HY_NT, HY_LDS_BYTES, st>>>(ha);
        else if (rbf) k_hyper<false, 0><<<grid_for(T, 1), HY_NT, HY_LDS_BYTES, st>>>(ha);
        else k_hyper<false, 1><<<grid_for(T, 1), HY_NT, HY_LDS_BYTES, st>>>(ha);
    } else {
    launch_alpha_refine(tv, b, w, st);
    if (with_hessian) {
        ProbP pp; pp.tv = tv; pp.Ainv = w.Ainv; pp.D2ss = w.D2ss; pp.P = w.P;
        launch_gemm(pp, T, ns, ns, st);
        HessArgs ha{tv, w.Ainv, w.P, w.D2ss, b->y_s, b->priors, w.scal, w.vecs, T};
        if (ns > REG_POINTS) {
            k_lg_hess_mv<<<dim3(ceil_div(ns, 4), T), 256, 0, st>>>(ha);
            LgMat am = lg_mat(w, w.Ainv, ns, b->n_s, nullptr, T);
            LgMatvecArgs mv{am, w.vecs + (size_t)V_BETA * w.vld, (size_t)NVEC * w.vld, w.vecs + (size_t)V_DELTA * w.vld, (size_t)NVEC * w.vld, 1.f};
            k_lg_matvec<<<dim3(ceil_div(ns, 4), T), 256, 0, st>>>(mv);
            LgHessTr ht{ha, w.lg_part, tms * tms, tms};
            k_lg_hess_tr<<<grid_for(T, tms * tms), 256, 0, st>>>(ht);
            k_lg_hess_fin<<<T, 64, 0, st>>>(ht);
        } else {
            k_hess<<<grid_for(T, 1), SMALL_NT, 0, st>>>(ha);
        }
    }
    launch_c(tv, b, w, st);
    ProbS ps; ps.tv = tv; ps.C = w.C; ps.D2qs = w.D2qs; ps.D2qq = w.D2qq; ps.S = w.S;
    launch_gemm(ps, T, nq, nq, st);
    // (reused inner stage: A^-1, alpha and the scalars of phi are in the workspace, info[] is written by the outer factor)
    OuterArgs oa{tv, w.C, w.S, b->y_s, b->y_q, w.vecs, w.scal, f_out, info, T, reuse_inner ? 1 : 0};
    rc = launch_outer_factor(oa, w, nq, st);
    if (rc) return rc;
    ProbOC po; po.tv = tv; po.Sinv = w.S; po.C = w.C; po.D2qs = w.D2qs; po.OC = w.OC; po.Wqs = w.Wqs; po.part = w.part_oc; po.ntiles = w.nt_oc; po.dirscale = dirscale;
    launch_gemm(po, T, nq, ns, st);
    ProbMA pm; pm.tv = tv; pm.C = w.C; pm.OC = w.OC; pm.D2ss = w.D2ss; pm.Wss = w.Wss; pm.part = w.part_ma; pm.ntiles = w.nt_ma; pm.dirscale = dirscale;
    launch_gemm(pm, T, ns, ns, st);
    SolveArgs sa{tv, w.scal, w.vecs, w.part_oc, w.part_ma, w.nt_oc, w.nt_ma, flags, g_phi_out, v_out, H_out, T, with_hessian ? 1 : 0};
    WqqArgs wq{tv, w.S, w.D2qq, w.Wqq, w.scal, dirscale, T, 0, sa};
    if (nq > REG_POINTS) {
        LgWqq lw{wq, w.lg_part, tmq * tmq, tmq};
        k_lg_wqq<<<grid_for(T, tmq * tmq), 256, 0, st>>>(lw);
        k_lg_wqq_fin<<<T, 64, 0, st>>>(lw);
        k_solve_v<<<T, 64, 0, st>>>(sa);
    } else {
        wq.do_solve = 1;   // g_out, v and w in the tail of the same workgroup
        k_wqq<<<grid_for(T, 1), SMALL_NT, 0, st>>>(wq);
    }
    if (corrscale != 0.f) {
        ProbMixed px; px.tv = tv; px.Ainv = w.Ainv; px.P = w.P; px.D2ss = w.D2ss; px.Wss = w.Wss; px.corrscale = corrscale;
        launch_gemm(px, T, ns, ns, st);
    }
    }
    if (dZ_s || dZ_q) {
        if (dZ_s && b->n_s) hipMemsetAsync(dZ_s, 0, (size_t)T * ns * d * sizeof(float), st);   // padded rows only exist in ragged batches
        if (dZ_q && b->n_q) hipMemsetAsync(dZ_q, 0, (size_t)T * nq * d * sizeof(float), st);
        ProbDZ<false> pzs; pzs.tv = tv; pzs.Wss = w.Wss; pzs.Wqs = w.Wqs; pzs.Wqq = w.Wqq; pzs.Zs = b->Z_s; pzs.Zq = b->Z_q; pzs.dZ = dZ_s; pzs.d = d;
        ProbDZ<true> pzq; pzq.tv = tv; pzq.Wss = w.Wss; pzq.Wqs = w.Wqs; pzq.Wqq = w.Wqq; pzq.Zs = b->Z_s; pzq.Zq = b->Z_q; pzq.dZ = dZ_q; pzq.d = d;
        // (both cotangents in ONE launch through gemm.h's select() hook, with the two functors behind a run-time switch, was
        // measured at 139.8 us against 62.6 + 56.4 for the two launches: dropped)
#if ADKF_VARIANT_DZ
        static const bool dz_optin = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dz), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                         (int)DZ_LDS_BYTES) == hipSuccess;
        if (dz_optin && ns == HY_N && nq == HY_N && !b->n_s && !b->n_q && tv.vec && (d % HY_N) == 0) {
            DzArgs da{w.Wss, w.Wqs, w.Wqq, b->Z_s, b->Z_q, dZ_s, dZ_q, d, T};
            k_dz<<<grid_for(T, 1), HY_NT, DZ_LDS_BYTES, st>>>(da);
        } else
#endif
        {
            if (dZ_s) launch_gemm(pzs, T, ns, d, st, x3_for(d));
            if (dZ_q) launch_gemm(pzq, T, nq, d, st, x3_for(d));
        }
    }
    if (w.w64) {
        // flagged (ill-conditioned) tasks, ONE launch at the very end: the factorisation-type stages (A^-1, alpha, P, the Hessian, C,
        // Sigma_q^-1, e, f_out) and then the cotangent algebra and dL/dZ, in float64, over what the kernels above wrote for them
        size_t lds_bytes;
        const Refine64Args ra = refine_args(tv, b, w, with_hessian, 2, f_out, info, nullptr, nullptr, nullptr, lds_bytes);
        Cot64Args ca{tv, b->Z_s, b->Z_q, dZ_s, dZ_q, d, w.vecs, w.scal, w.w64, w.w64_stride, r64_threshold(), T,
                     with_hessian ? 1 : 0, flags, dirscale, corrscale, g_phi_out, v_out, H_out, lds_bytes ? 1 : 0, r64_stop()};
        k_tail64<<<T, R64_NT, lds_bytes, st>>>(ra, ca);
    }
    LAUNCH_OK();
    return 0;
}


// ======================================================================================================================
// ARD (ADKF_BATCH_ARD): host side.  See ard.h for the formulation.
// ======================================================================================================================
struct ArdWs {
    float *mu, *ell, *Zt_s, *Zt_q, *G, *Gd_s, *Gd_q, *Gdot, *phi3, *pri3, *f3, *g3, *g3o, *S1, *gt, *coldot;
    float *c, *ut2, *wn, *Ddot, *Wdot, *adot, *S2;
    ArdFitState* fst; float *x, *g, *p, *xe, *ge, *S, *Y, *fe; int32_t* info3;
    ArdCgState* cst; float *cx, *cr, *cp, *cHp, *gout; int32_t* n_eff;
    size_t bytes;
};

ArdWs carve_ard(void* base, size_t off0, int T, int ns, int nq, int d) {
    ArdWs a;
    size_t off = off0;
    auto take = [&](size_t nfloat) { float* p = base ? reinterpret_cast<float*>(static_cast<char*>(base) + off) : nullptr; off += align_up((nfloat ? nfloat : 1) * sizeof(float)); return p; };
    const size_t Tz = (size_t)T, h = 2 + (size_t)d;
    a.mu = take(Tz * d); a.ell = take(Tz * d);
    a.Zt_s = take(Tz * ns * d); a.Zt_q = take(Tz * nq * d);
    a.G = take(Tz * ns * d); a.Gd_s = take(Tz * ns * d); a.Gd_q = take(Tz * nq * d); a.Gdot = take(Tz * ns * d);
    a.phi3 = take(Tz * 3); a.pri3 = take(Tz * 4); a.f3 = take(Tz); a.g3 = take(Tz * 3); a.g3o = take(Tz * 3);
    a.S1 = take(Tz * d); a.gt = take(Tz * h); a.coldot = take(Tz * d);
    a.c = take(Tz * d); a.ut2 = take(Tz * 2); a.wn = take(Tz * ns);
    a.Ddot = take(Tz * ns * ns); a.Wdot = take(Tz * ns * ns); a.adot = take(Tz * ns); a.S2 = take(Tz * d);
    a.fst = reinterpret_cast<ArdFitState*>(take(Tz * ((sizeof(ArdFitState) + 3) / 4)));
    a.x = take(Tz * h); a.g = take(Tz * h); a.p = take(Tz * h); a.xe = take(Tz * h); a.ge = take(Tz * h);
    a.S = take(Tz * ARD_M * h); a.Y = take(Tz * ARD_M * h); a.fe = take(Tz);
    a.info3 = reinterpret_cast<int32_t*>(take(Tz));
    a.cst = reinterpret_cast<ArdCgState*>(take(Tz * ((sizeof(ArdCgState) + 3) / 4)));
    a.cx = take(Tz * h); a.cr = take(Tz * h); a.cp = take(Tz * h); a.cHp = take(Tz * h); a.gout = take(Tz * h);
    a.n_eff = reinterpret_cast<int32_t*>(take(Tz));
    a.bytes = off;
    return a;
}

struct ArdCtx {
    const adkf_batch_t* b;
    adkf_batch_t bt;   // the scaled batch the non-ARD pipeline runs on
    Workspace w; ArdWs a; ArdView v;
    int T, ns, nq, d, h;
    hipStream_t st;
};

int ard_setup(const adkf_batch_t* b, void* ws, size_t ws_bytes, hipStream_t st, ArdCtx& c) {
    c.b = b; c.T = b->T; c.ns = b->ns_max; c.nq = b->nq_max; c.d = b->d; c.h = 2 + b->d; c.st = st;
    c.w = carve(ws, c.T, c.ns, c.nq, c.d);
    c.a = carve_ard(ws, c.w.bytes, c.T, c.ns, c.nq, c.d);
    if (ws_bytes < c.a.bytes) return ADKF_E_WORKSPACE;
    ArdView& v = c.v;
    v.T = c.T; v.d = c.d; v.h = c.h; v.ns_ld = c.ns; v.nq_ld = c.nq; v.n_s = b->n_s; v.n_q = b->n_q;
    v.Z_s = b->Z_s; v.Z_q = b->Z_q; v.Zt_s = c.a.Zt_s; v.Zt_q = c.a.Zt_q; v.mu = c.a.mu; v.ell = c.a.ell;
    v.phi3 = c.a.phi3; v.pri3 = c.a.pri3; v.priors = b->priors; v.f3 = c.a.f3; v.g3 = c.a.g3; v.S1 = c.a.S1; v.gt = c.a.gt;
    c.bt = *b;
    c.bt.Z_s = c.a.Zt_s; c.bt.Z_q = has_query(b) ? c.a.Zt_q : nullptr; c.bt.priors = c.a.pri3; c.bt.flags = 0;
    if (!(b->flags & ADKF_BATCH_REUSE_INNER))
        k_colmean<<<dim3(ceil_div(c.d, 64), c.T), 256, 0, st>>>(b->Z_s, b->n_s, c.ns, c.d, c.a.mu, c.T);
    hipMemsetAsync(c.w.mean, 0, sizeof(float) * (size_t)c.T * c.d, st);   // the scaled features are centred already (stage_dist parts bit 4)
    return 0;
}

// d f / d Z~_s for the weights in w.Wss (symmetric) -> out
void ard_dz_support(ArdCtx& c, const float* W, float* out, const int32_t* n_override = nullptr) {
    TaskView tv = make_tv(&c.bt, c.w, false);
    if (n_override) tv.n_s = n_override;
    ProbDZ<false> pz; pz.tv = tv; pz.Wss = W; pz.Wqs = nullptr; pz.Wqq = nullptr; pz.Zs = c.a.Zt_s; pz.Zq = nullptr; pz.dZ = out; pz.d = c.d;
    launch_gemm(pz, c.T, c.ns, c.d, c.st, x3_for(c.d));
}

// One evaluation of f_in and its gradient in the h raw parameters at x [T, h]; leaves Zt_s, D2ss, Ainv, alpha, the
// scalars, G = d f_in / d Z~, S1 and gt for x in the workspace.
int ard_eval(ArdCtx& c, const float* x, float* f, float* g, int32_t* info3) {
    hipStream_t st = c.st;
    k_ard_params<<<dim3(ceil_div(c.d, 256), c.T), 256, 0, st>>>(c.v, x);
    k_ard_scale<<<dim3(ceil_div(c.ns, 4), c.T), 256, 0, st>>>(c.v, c.b->Z_s, c.a.Zt_s, c.b->n_s, c.ns);
    int rc = stage_dist(&c.bt, c.w, false, st, 1 | 4);
    if (rc) return rc;
    InnerArgs ia = inner_args(&c.bt, c.w, c.a.phi3, info3);
    ia.f_out = c.a.f3; ia.g_out = c.a.g3;
    rc = launch_inner(ia, c.w, st);
    if (rc) return rc;
    TaskView tv = make_tv(&c.bt, c.w, false);
    const int win_tiles = std::max(1, std::min(64, c.ns * c.ns / 2048));
    WinArgs wa{tv, c.w.Ainv, c.w.D2ss, c.w.Wss, c.w.scal, c.T, win_tiles};
    k_win<<<grid_for(c.T, win_tiles), 256, 0, st>>>(wa);
    ard_dz_support(c, c.w.Wss, c.a.G);
    ArdColdot cd{c.a.Zt_s, c.a.G, c.b->n_s, c.ns, nullptr, nullptr, nullptr, 0, c.a.S1, c.d};
    k_ard_coldot<<<dim3(ceil_div(c.d, 64), c.T), 256, 0, st>>>(cd);
    ArdEvalFin ef{c.v, x, f, g, info3};
    k_ard_eval_fin<<<c.T, 256, 0, st>>>(ef);
    LAUNCH_OK();
    return 0;
}

// masked = true: sizes come from n_eff (0 for tasks whose CG has converged), so every kernel of the product skips them
ArdHvp ard_hvp_args(ArdCtx& c, const float* x, const float* u, float* Hu, const ArdCgState* cg, bool masked = false) {
    ArdHvp hv;
    hv.v = c.v; hv.tv = make_tv(&c.bt, c.w, false); hv.x = x; hv.u = u; hv.Hu = Hu;
    if (masked) { hv.v.n_s = c.a.n_eff; hv.tv.n_s = c.a.n_eff; }
    hv.c = c.a.c; hv.ut2 = c.a.ut2; hv.wn = c.a.wn; hv.D2 = c.w.D2ss; hv.Ainv = c.w.Ainv;
    hv.Ddot = c.a.Ddot; hv.X = c.w.P; hv.Wdot = c.a.Wdot; hv.adot = c.a.adot;
    hv.part = c.w.part_ma; hv.ntiles = c.w.nt_ma; hv.G = c.a.G; hv.Gdot = c.a.Gdot; hv.S2 = c.a.S2; hv.cg = cg;
    return hv;
}

// Everything of one Hessian-vector product up to Gdot' = 4 (rowsum(Wdot) . Z~ - Wdot Z~) (needed alone by the mixed term)
void ard_hvp_core(ArdCtx& c, const ArdHvp& hv) {
    hipStream_t st = c.st;
    k_ard_dir<<<dim3(ceil_div(c.d, 256), c.T), 256, 0, st>>>(hv);
    k_ard_wnorm<<<dim3(ceil_div(c.ns, 4), c.T), 256, 0, st>>>(hv);
    ProbArdDdot pd; pd.h = hv; launch_gemm(pd, c.T, c.ns, c.ns, st);
    ProbArdX px; px.h = hv; launch_gemm(px, c.T, c.ns, c.ns, st);
    k_ard_adot<<<dim3(ceil_div(c.ns, 4), c.T), 256, 0, st>>>(hv);
    ProbArdY py; py.h = hv; launch_gemm(py, c.T, c.ns, c.ns, st);
    ard_dz_support(c, c.a.Wdot, c.a.Gdot, hv.tv.n_s);
}

void ard_hvp(ArdCtx& c, const float* x, const float* u, float* Hu, const ArdCgState* cg) {
    ArdHvp hv = ard_hvp_args(c, x, u, Hu, cg, cg != nullptr);
    ard_hvp_core(c, hv);
    ArdColdot cd{c.a.Zt_s, c.a.Gdot, hv.tv.n_s, c.ns, nullptr, nullptr, nullptr, 0, c.a.S2, c.d};
    k_ard_coldot<<<dim3(ceil_div(c.d, 64), c.T), 256, 0, c.st>>>(cd);
    k_ard_hvp_fin<<<c.T, 256, 0, c.st>>>(hv);
}

__global__ void k_ard_expand_phi(const float* phi3, float* phi, int T, int h) {
    const int t = blockIdx.y, k = blockIdx.x * 256 + threadIdx.x;
    if (k < h) phi[(size_t)t * h + k] = phi3[t * 3 + (k < 2 ? k : 2)];
}

__global__ void k_ard_cg_info(const ArdCgState* cg, int32_t* info, int32_t* iters, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= T) return;
    if (iters) iters[t] = cg[t].iters;
    if (cg[t].breakdown && info[t] == 0) info[t] = 200000 + cg[t].iters + 1;   // H not positive definite along a CG direction
}

int ard_fit(const adkf_batch_t* b, float* phi, const adkf_fit_options_t* opt, float* f_final, float* gnorm, int32_t* n_evals,
            int32_t* info, void* ws, size_t ws_bytes, hipStream_t st) {
    ArdCtx c;
    adkf_batch_t b0 = *b; b0.flags &= ~ADKF_BATCH_REUSE_INNER;
    int rc = ard_setup(&b0, ws, ws_bytes, st, c);
    if (rc) return rc;
    ArdFitArgs fa;
    fa.T = c.T; fa.h = c.h; fa.max_evals = opt->max_evals; fa.exact_evals = opt->exact_evals; fa.gtol = opt->gtol; fa.ftol = opt->ftol;
    fa.st = c.a.fst; fa.x = c.a.x; fa.g = c.a.g; fa.p = c.a.p; fa.xe = c.a.xe; fa.ge = c.a.ge; fa.S = c.a.S; fa.Y = c.a.Y;
    fa.fe = c.a.fe; fa.info_eval = c.a.info3; fa.phi = phi; fa.f_final = f_final; fa.gnorm = gnorm; fa.nevals = n_evals; fa.info = info;
    k_ard_fit_begin<<<dim3(ceil_div(c.h, 256), c.T), 256, 0, st>>>(fa);
    if (opt->ev_start && hipEventRecord(static_cast<hipEvent_t>(opt->ev_start), st) != hipSuccess) return ADKF_E_LAUNCH;
    FitPoll poll(!opt->exact_evals, opt->max_evals, c.a.n_eff, st);   // n_eff[0] is only used by the CG of the hypergradient
    for (int e = 0; e < opt->max_evals; ++e) {
        rc = ard_eval(c, c.a.xe, c.a.fe, c.a.ge, c.a.info3);
        if (rc) return rc;
        k_ard_advance<<<c.T, 256, 0, st>>>(fa);
        if (poll.finished(e, c.a.fst, sizeof(ArdFitState), offsetof(ArdFitState, phase), c.T, st)) break;
    }
    if (opt->ev_stop && hipEventRecord(static_cast<hipEvent_t>(opt->ev_stop), st) != hipSuccess) return ADKF_E_LAUNCH;
    LAUNCH_OK();
    return 0;
}

// C = K_qs A^-1, predictive mean / variance (/ covariance) from the distances, A^-1 and scalars in the workspace
int predict_core(const adkf_batch_t* b, const Workspace& w, float* mean, float* var, float* cov, int32_t* info, hipStream_t st) {
    TaskView tv = make_tv(b, w, true);
    const int T = b->T;
    launch_alpha_refine(tv, b, w, st);
    launch_c(tv, b, w, st);
    launch_refine(tv, b, w, false, 1, nullptr, info, st);
    PredArgs pa{tv, w.C, w.D2qs, b->y_s, mean, var, w.scal, T};
    k_predict<<<grid_for(T, 1), 256, 0, st>>>(pa);
    if (cov) {
        hipMemsetAsync(cov, 0, (size_t)T * b->nq_max * b->nq_max * sizeof(float), st);
        ProbS ps; ps.tv = tv; ps.C = w.C; ps.D2qs = w.D2qs; ps.D2qq = w.D2qq; ps.S = cov;
        launch_gemm(ps, T, b->nq_max, b->nq_max, st);
    }
    LAUNCH_OK();
    return 0;
}

// The outer stages on the scaled batch: query scaling + distances, f_out, direct feature gradients, g_out (h entries).
int ard_outer(ArdCtx& c, const float* phi, int flags, float* f_out, int32_t* info, bool want_grads) {
    hipStream_t st = c.st;
    int rc;
    if (!(c.b->flags & ADKF_BATCH_REUSE_INNER)) {
        rc = ard_eval(c, phi, c.a.fe, c.a.ge, c.a.info3);
        if (rc) return rc;
        hipMemcpyAsync(info, c.a.info3, sizeof(int32_t) * (size_t)c.T, hipMemcpyDeviceToDevice, st);
    } else {
        hipMemsetAsync(info, 0, sizeof(int32_t) * (size_t)c.T, st);
    }
    k_ard_scale<<<dim3(ceil_div(c.nq, 4), c.T), 256, 0, st>>>(c.v, c.b->Z_q, c.a.Zt_q, c.b->n_q, c.nq);
    rc = stage_dist(&c.bt, c.w, true, st, 2 | 4);
    if (rc) return rc;
    if (!want_grads) return 0;
    adkf_batch_t bq = c.bt;
    bq.flags = ADKF_BATCH_REUSE_DIST | ADKF_BATCH_REUSE_INNER;
    int32_t* info_o = c.a.info3;   // outer factorisation status, merged below
    rc = outer_pipeline(&bq, c.w, c.a.phi3, flags & ADKF_IGNORE_DIRECT_GRAD, false, f_out, c.a.Gd_s, c.a.Gd_q, c.a.g3o, nullptr, nullptr, info_o, st);
    if (rc) return rc;
    ArdColdot cd{c.a.Zt_s, c.a.Gd_s, c.b->n_s, c.ns, c.a.Zt_q, c.a.Gd_q, c.b->n_q, c.nq, c.a.coldot, c.d};
    k_ard_coldot<<<dim3(ceil_div(c.d, 64), c.T), 256, 0, st>>>(cd);
    ArdGout go{c.v, phi, c.a.coldot, c.a.g3o, c.a.gout};
    k_ard_gout<<<dim3(ceil_div(c.d, 256), c.T), 256, 0, st>>>(go);
    LAUNCH_OK();
    return 0;
}

__global__ void k_merge_info(const int32_t* extra, int32_t* info, int T) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < T && info[t] == 0 && extra[t] != 0) info[t] = extra[t];
}

int ard_ift(const adkf_batch_t* b, const float* phi, int flags, bool with_hessian, int cg_maxiter, float cg_tol, float* f_out,
            float* dZ_s, float* dZ_q, float* g_phi_out, float* v_out, int32_t* cg_iters, int32_t* info, void* ws, size_t ws_bytes,
            hipStream_t st) {
    ArdCtx c;
    int rc = ard_setup(b, ws, ws_bytes, st, c);
    if (rc) return rc;
    if (b->flags & ADKF_BATCH_REUSE_INNER) k_ard_params<<<dim3(ceil_div(c.d, 256), c.T), 256, 0, st>>>(c.v, phi);
    rc = ard_outer(c, phi, flags, f_out, info, true);
    if (rc) return rc;
    k_merge_info<<<ceil_div(c.T, 64), 64, 0, st>>>(c.a.info3, info, c.T);
    const size_t hb = sizeof(float) * (size_t)c.T * c.h;
    if (g_phi_out) hipMemcpyAsync(g_phi_out, c.a.gout, hb, hipMemcpyDeviceToDevice, st);
    const bool correct = with_hessian && !(flags & ADKF_IGNORE_GRAD_CORRECTION);
    if (correct) {
        // (round 5, measured and dropped: CG preconditioned with the L-BFGS history the fit has just built at this point - two-loop
        // recursion per round - needed MORE rounds than plain CG at the C2 shapes, h = 258: 10.1 on average, 17 at most, against 9.0 / 11
        // (profiles/r05_bench_ard_pcg.json), and its step kernel took 19 us instead of 4.  Twenty evaluations of a 258-parameter fit do
        // not leave a useful picture of the curvature; the lengthscale prior already keeps cond(H) near 1e3.)
        ArdCg cg{c.T, c.h, cg_tol, c.a.cst, c.a.gout, c.a.cx, c.a.cr, c.a.cp, c.a.cHp, b->n_s, c.ns, c.a.n_eff};
        k_ard_cg_begin<<<c.T, 256, 0, st>>>(cg);
        FitPoll poll(true, cg_maxiter, c.a.info3, st);   // info3 was merged into info above; free as a counter now
        for (int it = 0; it < cg_maxiter; ++it) {
            ard_hvp(c, phi, c.a.cp, c.a.cHp, c.a.cst);
            k_ard_cg_step<<<c.T, 256, 0, st>>>(cg);
            if (poll.finished(it, c.a.cst, sizeof(ArdCgState), offsetof(ArdCgState, done), c.T, st, 1, 2)) break;   // plain CG needs 9 rounds on average, 11 at most at the C2 shapes
        }
        k_ard_cg_info<<<ceil_div(c.T, 64), 64, 0, st>>>(c.a.cst, info, cg_iters, c.T);
        if (v_out) hipMemcpyAsync(v_out, c.a.cx, hb, hipMemcpyDeviceToDevice, st);
        ArdHvp hv = ard_hvp_args(c, phi, c.a.cx, c.a.cHp, nullptr);
        ard_hvp_core(c, hv);   // Gdot'(v), c(v)
    } else {
        if (v_out) hipMemsetAsync(v_out, 0, hb, st);
        if (cg_iters) hipMemsetAsync(cg_iters, 0, sizeof(int32_t) * (size_t)c.T, st);
    }
    if (dZ_s) {
        ArdDzFin fs{c.v, c.a.Gd_s, correct ? c.a.Gdot : nullptr, c.a.G, c.a.c, correct ? 1.f : 0.f, dZ_s, b->n_s, c.ns};
        k_ard_dz_fin<<<dim3(ceil_div(c.d, 256), c.ns, c.T), 256, 0, st>>>(fs);
    }
    if (dZ_q) {
        ArdDzFin fq{c.v, c.a.Gd_q, nullptr, nullptr, nullptr, 0.f, dZ_q, b->n_q, c.nq};
        k_ard_dz_fin<<<dim3(ceil_div(c.d, 256), c.nq, c.T), 256, 0, st>>>(fq);
    }
    LAUNCH_OK();
    return 0;
}

inline bool is_ard(const adkf_batch_t* b) { return (b->flags & ADKF_BATCH_ARD) != 0; }

}  // namespace

extern "C" {

const char* adkf_last_hip_error(void) { return hipGetErrorString(g_last_hip_error); }

const char* adkf_version(void) { return "adkf_gp 0.1 (gfx950)"; }

int adkf_path_info(int32_t ns_max, int32_t nq_max) {
    if (ns_max <= 0 || nq_max < 0 || ns_max > MAX_POINTS || nq_max > MAX_POINTS) return ADKF_E_SIZE;
    int bits = 0;
    const int hi = ns_max > nq_max ? ns_max : nq_max;
    if (nq_max > 0 && use_fused_outer(ns_max, nq_max)) bits |= ADKF_PATH_FUSED_OUTER;
    if (hi > REG_POINTS) { bits |= ADKF_PATH_BLOCKED; if (lg_fused() && hi >= 4 * LB) bits |= ADKF_PATH_BLOCKED_FUSED; }
    if (carve(nullptr, 1, ns_max, nq_max > 0 ? nq_max : 1, 4).w64_stride != 0) bits |= ADKF_PATH_R64_REGION;
    if (refine64_lds_optin()) bits |= ADKF_PATH_R64_LDS;
    return bits;
}

int adkf_max_points(void) { return MAX_POINTS; }

size_t adkf_workspace_bytes(int32_t T, int32_t ns_max, int32_t nq_max, int32_t d) {
    if (T <= 0 || ns_max <= 0 || nq_max < 0 || d <= 0) return 0;
    return carve(nullptr, T, ns_max, nq_max, d).bytes;
}

// median heuristic (+ optionally a4 in the same launch); returns true through *fused when init was applied
static int median_core(const adkf_batch_t* b, const Workspace& w, float* l0, const InitArgs& init, bool* fused, hipStream_t st) {
    int rc = stage_dist(b, w, has_query(b), st);
    if (rc) return rc;
    *fused = b->ns_max <= 256;
    if (b->ns_max <= 128) k_median<512, 16><<<grid_for(b->T, 1), 512, 0, st>>>(w.D2ss, b->n_s, b->ns_max, l0, b->T, init);
    else if (b->ns_max <= 256) k_median<1024, 32><<<grid_for(b->T, 1), 1024, 0, st>>>(w.D2ss, b->n_s, b->ns_max, l0, b->T, init);
    else {
        LgMedian lm{w.D2ss, b->n_s, b->ns_max, l0, b->T, reinterpret_cast<uint32_t*>(w.lg_med), w.lg_med + b->T, w.lg_med + 2 * (size_t)b->T};
        hipMemsetAsync(lm.hist, 0, sizeof(int) * 256 * (size_t)b->T, st);
        const int rows_blocks = ceil_div(b->ns_max, 16) < 64 ? ceil_div(b->ns_max, 16) : 64;
        for (int pass = 0; pass < 4; ++pass) {
            k_lg_med_hist<<<dim3(rows_blocks, b->T), 256, 0, st>>>(lm, pass);
            k_lg_med_pick<<<ceil_div(b->T, 64), 64, 0, st>>>(lm, pass);
        }
    }
    LAUNCH_OK();
    return 0;
}

int adkf_median_lengthscale(const adkf_batch_t* b, float* l0, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_batch(b, false);
    if (rc) return rc;
    if (!l0 || !ws) return ADKF_E_BADARG;
    Workspace w = carve_for(b, ws);
    if (ws_bytes < w.bytes) return ADKF_E_WORKSPACE;
    bool fused;
    return median_core(b, w, l0, InitArgs{0, 0, nullptr, nullptr}, &fused, static_cast<hipStream_t>(stream));
}

int adkf_init_params(const adkf_batch_t* b, int32_t use_numeric_labels, int32_t use_lengthscale_prior, float* phi,
                     float* priors, float* l0, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_batch(b, false);
    if (rc) return rc;
    if (!phi || !priors || !ws) return ADKF_E_BADARG;
    Workspace w = carve_for(b, ws);
    if (ws_bytes < w.bytes) return ADKF_E_WORKSPACE;
    float* l0p = l0 ? l0 : w.l0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    InitArgs init{use_numeric_labels, use_lengthscale_prior, phi, priors};
    ArdWs a{};
    if (is_ard(b)) {  // every lengthscale starts at the median heuristic (adaptive_dkt.py:101)
        a = carve_ard(ws, w.bytes, b->T, b->ns_max, b->nq_max, b->d);
        if (ws_bytes < a.bytes) return ADKF_E_WORKSPACE;
        init.phi = a.phi3;
    }
    bool fused = false;
    rc = median_core(b, w, l0p, init, &fused, st);
    if (rc) return rc;
    if (!fused) k_init_params<<<ceil_div(b->T, 64), 64, 0, st>>>(l0p, b->T, init);
    if (is_ard(b)) k_ard_expand_phi<<<dim3(ceil_div(2 + b->d, 256), b->T), 256, 0, st>>>(a.phi3, phi, b->T, 2 + b->d);
    LAUNCH_OK();
    return 0;
}

int adkf_mll_value_grad(const adkf_batch_t* b, const float* phi, float* f_in, float* g_phi, float* dZ_s, int32_t* info,
                        void* ws, size_t ws_bytes, void* stream) {
    int rc = check_batch(b, false);
    if (rc) return rc;
    if (!phi || !f_in || !info || !ws || !b->y_s || !b->priors) return ADKF_E_BADARG;
    if (is_ard(b)) {
        ArdCtx c;
        adkf_batch_t b0 = *b; b0.flags &= ~ADKF_BATCH_REUSE_INNER;
        rc = ard_setup(&b0, ws, ws_bytes, static_cast<hipStream_t>(stream), c);
        if (rc) return rc;
        rc = ard_eval(c, phi, f_in, g_phi ? g_phi : c.a.ge, info);
        if (rc) return rc;
        if (dZ_s) {
            ArdDzFin fs{c.v, c.a.G, nullptr, nullptr, nullptr, 0.f, dZ_s, b->n_s, c.ns};
            k_ard_dz_fin<<<dim3(ceil_div(c.d, 256), c.ns, c.T), 256, 0, c.st>>>(fs);
            LAUNCH_OK();
        }
        return 0;
    }
    Workspace w = carve_for(b, ws);
    if (ws_bytes < w.bytes) return ADKF_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = stage_dist(b, w, has_query(b), st);
    if (rc) return rc;
    InnerArgs ia = inner_args(b, w, const_cast<float*>(phi), info);
    ia.f_out = f_in; ia.g_out = g_phi;
    rc = launch_inner(ia, w, st);
    if (rc) return rc;
    launch_refine(make_tv(b, w, false), b, w, false, 0, nullptr, info, st, f_in, g_phi, nullptr);
    if (dZ_s) {
        TaskView tv = make_tv(b, w, false);
        const int win_tiles = std::max(1, std::min(64, b->ns_max * b->ns_max / 2048));
        WinArgs wa{tv, w.Ainv, w.D2ss, w.Wss, w.scal, b->T, win_tiles};
        k_win<<<grid_for(b->T, win_tiles), 256, 0, st>>>(wa);
        hipMemsetAsync(dZ_s, 0, (size_t)b->T * b->ns_max * b->d * sizeof(float), st);
        ProbDZ<false> pz; pz.tv = tv; pz.Wss = w.Wss; pz.Wqs = nullptr; pz.Wqq = nullptr; pz.Zs = b->Z_s; pz.Zq = nullptr; pz.dZ = dZ_s; pz.d = b->d;
        launch_gemm(pz, b->T, b->ns_max, b->d, st, x3_for(b->d));
        LAUNCH_OK();
    }
    return 0;
}

int adkf_fit(const adkf_batch_t* b, float* phi, const adkf_fit_options_t* opt, float* f_final, float* gnorm,
             int32_t* n_evals, int32_t* info, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_batch(b, false);
    if (rc) return rc;
    if (!phi || !opt || !info || !ws || !b->y_s || !b->priors || opt->max_evals < 2) return ADKF_E_BADARG;
    if (is_ard(b)) return ard_fit(b, phi, opt, f_final, gnorm, n_evals, info, ws, ws_bytes, static_cast<hipStream_t>(stream));
    Workspace w = carve_for(b, ws);
    if (ws_bytes < w.bytes) return ADKF_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = stage_dist(b, w, has_query(b), st);
    if (rc) return rc;
    InnerArgs ia = inner_args(b, w, phi, info);
    ia.f_out = f_final; ia.gnorm_out = gnorm; ia.nevals_out = n_evals;
    ia.max_evals = opt->max_evals; ia.exact_evals = opt->exact_evals; ia.gtol = opt->gtol; ia.ftol = opt->ftol;
    if (opt->ev_start && hipEventRecord(static_cast<hipEvent_t>(opt->ev_start), st) != hipSuccess) return ADKF_E_LAUNCH;
    rc = launch_inner(ia, w, st);
    if (opt->ev_stop && hipEventRecord(static_cast<hipEvent_t>(opt->ev_stop), st) != hipSuccess) return ADKF_E_LAUNCH;
    if (rc) return rc;
    // ill-conditioned tasks: float64 value at phi* - unless the caller says that the next call on this workspace redoes it anyway
    if (!(b->flags & ADKF_BATCH_DEFER_REFINE)) launch_refine(make_tv(b, w, false), b, w, false, 0, nullptr, info, st, f_final, nullptr, gnorm);
    LAUNCH_OK();
    return 0;
}

int adkf_predict(const adkf_batch_t* b, const float* phi, float* mean, float* var, float* cov, int32_t* info, void* ws,
                 size_t ws_bytes, void* stream) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    if (!phi || !mean || !info || !ws || !b->y_s || !b->priors) return ADKF_E_BADARG;
    if (is_ard(b)) {
        ArdCtx c;
        rc = ard_setup(b, ws, ws_bytes, static_cast<hipStream_t>(stream), c);
        if (rc) return rc;
        if (b->flags & ADKF_BATCH_REUSE_INNER) k_ard_params<<<dim3(ceil_div(c.d, 256), c.T), 256, 0, c.st>>>(c.v, phi);
        rc = ard_outer(c, phi, 0, nullptr, info, false);
        if (rc) return rc;
        adkf_batch_t bq = c.bt;
        return predict_core(&bq, c.w, mean, var, cov, info, c.st);
    }
    Workspace w = carve_for(b, ws);
    if (ws_bytes < w.bytes) return ADKF_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = stage_dist(b, w, true, st);
    if (rc) return rc;
    if (b->flags & ADKF_BATCH_REUSE_INNER) {
        hipMemsetAsync(info, 0, sizeof(int32_t) * (size_t)b->T, st);
    } else {
        InnerArgs ia = inner_args(b, w, const_cast<float*>(phi), info);
        rc = launch_inner(ia, w, st);
        if (rc) return rc;
    }
    return predict_core(b, w, mean, var, cov, info, st);
}

int adkf_outer_nll_value_grad(const adkf_batch_t* b, const float* phi, float* f_out, float* g_phi, float* dZ_s,
                              float* dZ_q, int32_t* info, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    if (!phi || !f_out || !info || !ws || !b->y_s || !b->y_q || !b->priors) return ADKF_E_BADARG;
    if (is_ard(b))
        return ard_ift(b, phi, 0, false, 0, 0.f, f_out, dZ_s, dZ_q, g_phi, nullptr, nullptr, info, ws, ws_bytes, static_cast<hipStream_t>(stream));
    Workspace w = carve_for(b, ws);
    if (ws_bytes < w.bytes) return ADKF_E_WORKSPACE;
    return outer_pipeline(b, w, phi, 0, false, f_out, dZ_s, dZ_q, g_phi, nullptr, nullptr, info, static_cast<hipStream_t>(stream));
}

int adkf_ift_hypergrad(const adkf_batch_t* b, const float* phi, int32_t flags, float* f_out, float* dZ_s, float* dZ_q,
                       float* g_phi_out, float* v, float* H, int32_t* info, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    if (!phi || !f_out || !dZ_s || !dZ_q || !info || !ws || !b->y_s || !b->y_q || !b->priors) return ADKF_E_BADARG;
    if (is_ard(b)) {
        if (H) return ADKF_E_BADARG;   // the h x h Hessian is never formed: the system is solved by conjugate gradients
        return ard_ift(b, phi, flags, true, ADKF_CG_DEFAULT_MAXITER, ADKF_CG_DEFAULT_TOL, f_out, dZ_s, dZ_q, g_phi_out, v, nullptr, info,
                       ws, ws_bytes, static_cast<hipStream_t>(stream));
    }
    Workspace w = carve_for(b, ws);
    if (ws_bytes < w.bytes) return ADKF_E_WORKSPACE;
    return outer_pipeline(b, w, phi, flags, true, f_out, dZ_s, dZ_q, g_phi_out, v, H, info, static_cast<hipStream_t>(stream));
}

int adkf_double_path_tasks(const adkf_batch_t* b, int32_t* flagged, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    if (!flagged || !ws || is_ard(b)) return ADKF_E_BADARG;
    Workspace w = carve_for(b, ws);
    if (ws_bytes < w.bytes) return ADKF_E_WORKSPACE;
    (void)hipGetLastError();
    k_double_path_tasks<<<ceil_div(b->T, 256), 256, 0, static_cast<hipStream_t>(stream)>>>(w.scal, b->ns_max, b->nq_max, w.w64 ? r64_threshold() : INFINITY, b->T, flagged);
    LAUNCH_OK();
    return 0;
}

size_t adkf_workspace_bytes_ard(int32_t T, int32_t ns_max, int32_t nq_max, int32_t d) {
    if (T <= 0 || ns_max <= 0 || nq_max < 0 || d <= 0) return 0;
    return carve_ard(nullptr, carve(nullptr, T, ns_max, nq_max, d).bytes, T, ns_max, nq_max, d).bytes;
}

int adkf_ift_hypergrad_cg(const adkf_batch_t* b, const float* phi, int32_t flags, int32_t cg_maxiter, float cg_tol, float* f_out,
                          float* dZ_s, float* dZ_q, float* g_phi_out, float* v, int32_t* cg_iters, int32_t* info, void* ws,
                          size_t ws_bytes, void* stream) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    if (!is_ard(b) || cg_maxiter < 1 || !(cg_tol > 0.f)) return ADKF_E_BADARG;
    if (!phi || !f_out || !dZ_s || !dZ_q || !info || !ws || !b->y_s || !b->y_q || !b->priors) return ADKF_E_BADARG;
    return ard_ift(b, phi, flags, true, cg_maxiter, cg_tol, f_out, dZ_s, dZ_q, g_phi_out, v, cg_iters, info, ws, ws_bytes,
                   static_cast<hipStream_t>(stream));
}

namespace {
// fills the per-edge-type table of MsgArgs; returns the total number of edges, or -1 for a bad argument
long msg_table(MsgArgs& m, const adkf_msg_et_t* ets, int n_et, bool backward) {
    long e_all = 0;
    int splits = 0;
    for (int q = 0; q < n_et; ++q) {
        const adkf_msg_et_t& s = ets[q];
        if (s.E < 0 || !s.W || (s.E > 0 && (!s.src || !s.tgt)) || (!backward && !s.bias) || (backward && (!s.dW || !s.db))) return -1;
        MsgEt& et = msg_row(m, q);
        et.src = s.src; et.tgt = s.tgt; et.W = s.W; et.bias = s.bias; et.dW = s.dW; et.db = s.db; et.E = s.E;
        et.e_off = (int)e_all; et.tile0 = 0; et.split0 = splits; et.chunk = s.E > 0 ? msg_chunk(s.E) : 1;
        splits += msg_nsplit(s.E);
        e_all += s.E;
        if (e_all > 0x7fffffffL) return -1;
    }
    m.n_et = n_et; m.nsplit_all = splits;
    return e_all;
}
int msg_tiles(MsgArgs& m, int n_cols) {   // lays the edge types' tiles side by side for a launch whose result has n_cols columns
    int total = 0;
    for (int q = 0; q < m.n_et; ++q) { msg_row(m, q).tile0 = total; total += ceil_div(msg_row(m, q).E, GT) * ceil_div(n_cols, GT); }
    return total;
}
}  // namespace

int adkf_msg_forward(const float* x, const adkf_msg_et_t* ets, int32_t n_et, int32_t H, int32_t in, int32_t out, float* msgs, void* stream) {
    (void)hipGetLastError();
    if (!x || !ets || !msgs || n_et <= 0 || n_et > MSG_MAX_ET || H <= 0 || in <= 0 || out <= 0) return ADKF_E_BADARG;
    MsgArgs m{};
    m.x = x; m.msgs = msgs; m.H = H; m.in = in; m.out = out;
    if (msg_table(m, ets, n_et, false) < 0) return ADKF_E_BADARG;
    m.vec = (in % 4 == 0) && (out % 4 == 0) && aligned16(x) && aligned16(msgs);
    for (int q = 0; q < n_et; ++q) m.vec = m.vec && aligned16(ets[q].W);
    const int total = msg_tiles(m, out);
    if (total == 0) return 0;
    ProbMsgFwd p; p.m = m;
    k_bgemm<ProbMsgFwd, GT><<<grid_for(H, total), 256, 0, static_cast<hipStream_t>(stream)>>>(p, H, 1, total);
    LAUNCH_OK();
    return 0;
}

size_t adkf_msg_backward_scratch_bytes(const adkf_msg_et_t* ets, int32_t n_et, int32_t H, int32_t in, int32_t out) {
    if (!ets || n_et <= 0 || n_et > MSG_MAX_ET || H <= 0 || in <= 0 || out <= 0) return 0;
    size_t splits = 0;
    for (int q = 0; q < n_et; ++q) splits += (size_t)msg_nsplit(ets[q].E);
    return sizeof(float) * splits * msg_part_stride(H, in, out);
}

int adkf_msg_backward(const float* x, const adkf_msg_et_t* ets, int32_t n_et, int32_t H, int32_t in, int32_t out, const float* msgs,
                      const float* d_msgs, const int64_t* perm_src, const int64_t* rowptr_src, const int64_t* perm_tgt,
                      const int64_t* rowptr_tgt, int32_t V, float* dcat, float* dx, void* scratch, size_t scratch_bytes, void* stream) {
    (void)hipGetLastError();
    if (!x || !ets || !d_msgs || !dcat || !dx || !perm_src || !rowptr_src || !perm_tgt || !rowptr_tgt) return ADKF_E_BADARG;   // (msgs may be null: d_msgs already masked)
    if (n_et <= 0 || n_et > MSG_MAX_ET || H <= 0 || in <= 0 || out <= 0 || V <= 0) return ADKF_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    MsgArgs m{};
    m.x = x; m.msgs = const_cast<float*>(msgs); m.d_msgs = d_msgs; m.dcat = dcat; m.part = static_cast<float*>(scratch);
    m.H = H; m.in = in; m.out = out;
    const long e_all = msg_table(m, ets, n_et, true);
    if (e_all < 0) return ADKF_E_BADARG;
    if (m.nsplit_all > 0 && (!scratch || scratch_bytes < adkf_msg_backward_scratch_bytes(ets, n_et, H, in, out))) return ADKF_E_WORKSPACE;
    m.vec = (in % 4 == 0) && (out % 4 == 0) && aligned16(x) && (!msgs || aligned16(msgs)) && aligned16(d_msgs);
    for (int q = 0; q < n_et; ++q) m.vec = m.vec && aligned16(ets[q].W);
    const int total = msg_tiles(m, 2 * in);
    if (total > 0) {
        ProbMsgBwdX px; px.m = m;
        k_bgemm<ProbMsgBwdX, GT><<<grid_for(H, total), 256, 0, st>>>(px, H, 1, total);
        ProbMsgBwdW pw; pw.m = m;
        launch_gemm(pw, H * m.nsplit_all, 2 * in, out, st);
        k_msg_dbias<<<dim3(ceil_div(H * out, 64), m.nsplit_all), 256, 0, st>>>(m);
    }
    // d W / d b of every edge type (exact zeros where it has no edges), then d x gathered over each node's edge lists
    k_msg_reduce<<<dim3(ceil_div(H * 2 * in * out + H * out, 256), n_et), 256, 0, st>>>(m);
    MsgDxArgs da{dcat, perm_src, rowptr_src, perm_tgt, rowptr_tgt, dx, V, H, in};
    const long n = (long)V * H * in;
    k_msg_dx<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(da);
    LAUNCH_OK();
    return 0;
}

int adkf_readout_pool(const float* s_mean, const float* v_mean, const float* s_sum, const float* v_sum, const float* emb,
                      const int64_t* perm, const int64_t* rowptr, int32_t V, int32_t G, int32_t nh, int32_t hd, int32_t D,
                      float* w_mean, float* w_sum, float* g_mean, float* g_sum, float* g_max, int32_t* argmax, void* stream) {
    (void)hipGetLastError();
    if (!s_mean || !v_mean || !s_sum || !v_sum || !emb || !perm || !rowptr || !w_mean || !w_sum || !g_mean || !g_sum || !g_max || !argmax)
        return ADKF_E_BADARG;
    if (V < 0 || G <= 0 || nh <= 0 || nh > READOUT_MAX_HEADS || hd <= 0 || D <= 0) return ADKF_E_BADARG;
    ReadoutArgs a{};
    a.s_mean = s_mean; a.v_mean = v_mean; a.s_sum = s_sum; a.v_sum = v_sum; a.emb = emb; a.perm = perm; a.rowptr = rowptr;
    a.w_mean = w_mean; a.w_sum = w_sum; a.g_mean = g_mean; a.g_sum = g_sum; a.g_max = g_max; a.argmax = argmax;
    a.V = V; a.G = G; a.nh = nh; a.hd = hd; a.D = D;
    k_readout_fwd<<<G, 256, 0, static_cast<hipStream_t>(stream)>>>(a);
    LAUNCH_OK();
    return 0;
}

int adkf_readout_pool_backward(const float* v_mean, const float* v_sum, const float* w_mean, const float* w_sum, const float* g_mean,
                               const int32_t* argmax, const int64_t* node_to_graph, const float* dg_mean, const float* dg_sum,
                               const float* dg_max, int32_t V, int32_t G, int32_t nh, int32_t hd, int32_t D, float* d_s_mean,
                               float* d_v_mean, float* d_s_sum, float* d_v_sum, float* d_emb, void* stream) {
    (void)hipGetLastError();
    if (!v_mean || !v_sum || !w_mean || !w_sum || !g_mean || !argmax || !node_to_graph || !dg_mean || !dg_sum || !dg_max || !d_s_mean ||
        !d_v_mean || !d_s_sum || !d_v_sum || !d_emb)
        return ADKF_E_BADARG;
    if (V < 0 || G <= 0 || nh <= 0 || hd <= 0 || D <= 0) return ADKF_E_BADARG;
    if (V == 0) return 0;
    ReadoutArgs a{};
    a.v_mean = v_mean; a.v_sum = v_sum; a.w_mean = const_cast<float*>(w_mean); a.w_sum = const_cast<float*>(w_sum);
    a.g_mean = const_cast<float*>(g_mean); a.argmax = const_cast<int32_t*>(argmax); a.n2g = node_to_graph;
    a.dg_mean = dg_mean; a.dg_sum = dg_sum; a.dg_max = dg_max;
    a.d_s_mean = d_s_mean; a.d_v_mean = d_v_mean; a.d_s_sum = d_s_sum; a.d_v_sum = d_v_sum; a.d_emb = d_emb;
    a.V = V; a.G = G; a.nh = nh; a.hd = hd; a.D = D;
    k_readout_bwd<<<V, 256, 0, static_cast<hipStream_t>(stream)>>>(a);
    LAUNCH_OK();
    return 0;
}

int adkf_readout_pool_hidden(const float* s_mean, const float* h_mean, const float* s_sum, const float* h_sum, int32_t ldh, const float* emb,
                             const int64_t* perm, const int64_t* rowptr, int32_t V, int32_t G, int32_t nh, int32_t K, int32_t D,
                             float* w_mean, float* w_sum, float* p_mean, float* p_sum, float* wtot_mean, float* wtot_sum, float* g_max,
                             int32_t* argmax, void* stream) {
    (void)hipGetLastError();
    if (!s_mean || !h_mean || !s_sum || !h_sum || !emb || !perm || !rowptr || !w_mean || !w_sum || !p_mean || !p_sum || !wtot_mean || !wtot_sum ||
        !g_max || !argmax)
        return ADKF_E_BADARG;
    if (V < 0 || G <= 0 || nh <= 0 || nh > READOUT_MAX_HEADS || K <= 0 || K > 256 * READOUT_KJ_MAX || ldh < K || D <= 0 || D > READOUT_MAX_D) return ADKF_E_BADARG;
    ReadoutHArgs a{};
    a.s_mean = s_mean; a.h_mean = h_mean; a.s_sum = s_sum; a.h_sum = h_sum; a.ldh = ldh; a.emb = emb; a.perm = perm; a.rowptr = rowptr;
    a.w_mean = w_mean; a.w_sum = w_sum; a.p_mean = p_mean; a.p_sum = p_sum; a.wtot_mean = wtot_mean; a.wtot_sum = wtot_sum;
    a.g_max = g_max; a.argmax = argmax; a.V = V; a.G = G; a.nh = nh; a.K = K; a.D = D;
    launch_readout_h(a, false, static_cast<hipStream_t>(stream));
    LAUNCH_OK();
    return 0;
}

int adkf_readout_pool_hidden_backward(const float* h_mean, const float* h_sum, int32_t ldh, const float* w_mean, const float* w_sum,
                                      const int32_t* argmax, const int64_t* perm, const int64_t* rowptr, const float* dp_mean,
                                      const float* dp_sum, const float* dwtot_sum, const float* dg_max, int32_t V, int32_t G, int32_t nh,
                                      int32_t K, int32_t D, float* d_s_mean, float* d_h_mean, float* d_s_sum, float* d_h_sum, float* d_emb,
                                      void* stream) {
    (void)hipGetLastError();
    if (!h_mean || !h_sum || !w_mean || !w_sum || !argmax || !perm || !rowptr || !dp_mean || !dp_sum || !dwtot_sum || !dg_max || !d_s_mean ||
        !d_h_mean || !d_s_sum || !d_h_sum || !d_emb)
        return ADKF_E_BADARG;
    if (V < 0 || G <= 0 || nh <= 0 || nh > READOUT_MAX_HEADS || K <= 0 || K > 256 * READOUT_KJ_MAX || ldh < K || D <= 0 || D > READOUT_MAX_D) return ADKF_E_BADARG;
    if (V == 0) return 0;
    ReadoutHArgs a{};
    a.h_mean = h_mean; a.h_sum = h_sum; a.ldh = ldh; a.w_mean = const_cast<float*>(w_mean); a.w_sum = const_cast<float*>(w_sum);
    a.argmax = const_cast<int32_t*>(argmax); a.perm = perm; a.rowptr = rowptr;
    a.dp_mean = dp_mean; a.dp_sum = dp_sum; a.dwtot_sum = dwtot_sum; a.dg_max = dg_max;
    a.d_s_mean = d_s_mean; a.d_h_mean = d_h_mean; a.d_s_sum = d_s_sum; a.d_h_sum = d_h_sum; a.d_emb = d_emb;
    a.V = V; a.G = G; a.nh = nh; a.K = K; a.D = D;
    launch_readout_h(a, true, static_cast<hipStream_t>(stream));
    LAUNCH_OK();
    return 0;
}

int adkf_pna_aggregate(const float* msgs, const int64_t* perm, const int64_t* rowptr, int32_t V, int32_t H, int32_t m, float* agg,
                       int32_t* argmax, void* stream) {
    (void)hipGetLastError();
    if (!msgs || !perm || !rowptr || !agg || !argmax || V <= 0 || H <= 0 || m <= 0) return ADKF_E_BADARG;
    PnaArgs a{msgs, perm, rowptr, agg, argmax, nullptr, nullptr, V, H, m};
    k_pna_fwd<<<V, 256, 0, static_cast<hipStream_t>(stream)>>>(a);
    LAUNCH_OK();
    return 0;
}

int adkf_pna_aggregate_backward(const float* msgs, const int64_t* perm, const int64_t* rowptr, const float* agg, const int32_t* argmax,
                                const float* d_agg, int32_t V, int32_t H, int32_t m, float* d_msgs, void* stream) {
    (void)hipGetLastError();
    if (!msgs || !perm || !rowptr || !agg || !argmax || !d_agg || !d_msgs || V <= 0 || H <= 0 || m <= 0) return ADKF_E_BADARG;
    PnaArgs a{msgs, perm, rowptr, const_cast<float*>(agg), const_cast<int32_t*>(argmax), d_agg, d_msgs, V, H, m, 0};
    k_pna_bwd<<<V, 256, 0, static_cast<hipStream_t>(stream)>>>(a);
    LAUNCH_OK();
    return 0;
}

int adkf_pna_aggregate_backward_relu(const float* msgs, const int64_t* perm, const int64_t* rowptr, const float* agg, const int32_t* argmax,
                                     const float* d_agg, int32_t V, int32_t H, int32_t m, float* d_pre, void* stream) {
    (void)hipGetLastError();
    if (!msgs || !perm || !rowptr || !agg || !argmax || !d_agg || !d_pre || V <= 0 || H <= 0 || m <= 0) return ADKF_E_BADARG;
    PnaArgs a{msgs, perm, rowptr, const_cast<float*>(agg), const_cast<int32_t*>(argmax), d_agg, d_pre, V, H, m, 1};
    k_pna_bwd<<<V, 256, 0, static_cast<hipStream_t>(stream)>>>(a);
    LAUNCH_OK();
    return 0;
}

int adkf_block_combine(const float* p, const float* x, const float* amp, const float* att, const float* bias, const float* alpha,
                       const float* gamma, const float* beta, float eps, int32_t V, int32_t hid, float* x1, float* h, float* mu,
                       float* rstd, void* stream) {
    (void)hipGetLastError();
    if (!p || !x || !amp || !att || !bias || !alpha || !gamma || !beta || !x1 || !h || !mu || !rstd) return ADKF_E_BADARG;
    if (V <= 0 || hid <= 0 || (hid % 64) || hid > 64 * BLK_MAXC) return ADKF_E_SIZE;
    BlockArgs a{};
    a.p = p; a.x = x; a.amp = amp; a.att = att; a.bias = bias; a.alpha = alpha; a.gamma = gamma; a.beta = beta;
    a.x1 = x1; a.h = h; a.mu = mu; a.rstd = rstd; a.eps = eps; a.V = V; a.hid = hid;
    int grid = ceil_div(V, 4);
    grid = grid > 16384 ? 16384 : grid;
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (hid / 64) {
        case 1: k_block_fwd<1><<<grid, 256, 0, st>>>(a); break;
        case 2: k_block_fwd<2><<<grid, 256, 0, st>>>(a); break;
        case 3: k_block_fwd<3><<<grid, 256, 0, st>>>(a); break;
        default: k_block_fwd<4><<<grid, 256, 0, st>>>(a); break;
    }
    LAUNCH_OK();
    return 0;
}

size_t adkf_block_combine_scratch_bytes(int32_t V, int32_t hid) {
    if (V <= 0 || hid <= 0) return 0;
    return sizeof(float) * (size_t)ceil_div(V, BLK_ROWS) * (3 * (size_t)hid + 1);
}

int adkf_block_combine_backward(const float* p, const float* x1, const float* amp, const float* att, const float* bias,
                                const float* alpha, const float* gamma, const float* mu, const float* rstd, const float* g_x1,
                                const float* g_h, int32_t V, int32_t hid, float* d_p, float* d_x, float* d_bias, float* d_alpha,
                                float* d_gamma, float* d_beta, void* scratch, size_t scratch_bytes, void* stream) {
    (void)hipGetLastError();
    if (!p || !x1 || !amp || !att || !bias || !alpha || !gamma || !mu || !rstd || !g_x1 || !g_h || !d_p || !d_x || !d_bias || !d_alpha ||
        !d_gamma || !d_beta || !scratch)
        return ADKF_E_BADARG;
    if (V <= 0 || hid <= 0 || (hid % 64) || hid > 64 * BLK_MAXC) return ADKF_E_SIZE;
    if (scratch_bytes < adkf_block_combine_scratch_bytes(V, hid)) return ADKF_E_WORKSPACE;
    BlockArgs a{};
    a.p = p; a.x1 = const_cast<float*>(x1); a.amp = amp; a.att = att; a.bias = bias; a.alpha = alpha; a.gamma = gamma;
    a.mu = const_cast<float*>(mu); a.rstd = const_cast<float*>(rstd); a.g_x1 = g_x1; a.g_h = g_h; a.d_p = d_p; a.d_x = d_x;
    a.part = static_cast<float*>(scratch); a.V = V; a.hid = hid;
    const int nwg = ceil_div(V, BLK_ROWS);
    hipStream_t st = static_cast<hipStream_t>(stream);
    switch (hid / 64) {
        case 1: k_block_bwd<1><<<nwg, 256, 0, st>>>(a); break;
        case 2: k_block_bwd<2><<<nwg, 256, 0, st>>>(a); break;
        case 3: k_block_bwd<3><<<nwg, 256, 0, st>>>(a); break;
        default: k_block_bwd<4><<<nwg, 256, 0, st>>>(a); break;
    }
    const int n = 3 * hid + 1;
    k_block_reduce<<<ceil_div(n, 64), 64, 0, st>>>(a.part, nwg, n, d_bias, d_gamma, d_beta, d_alpha, hid);
    LAUNCH_OK();
    return 0;
}

int adkf_split_planes(const float* x, uint16_t* planes, int64_t rows, int64_t K, void* stream) {
    (void)hipGetLastError();
    if (!x || !planes || rows <= 0 || K <= 0 || (K & 1)) return ADKF_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(x) & 7) || (reinterpret_cast<uintptr_t>(planes) & 15) || ((rows * K) & 7)) return ADKF_E_BADARG;
    const size_t pairs = (size_t)rows * (size_t)K / 2;
    if (pairs > (size_t)0x7fffffff * 256) return ADKF_E_SIZE;
    k_split3<<<(unsigned)((pairs + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(x, planes, pairs, (size_t)rows * (size_t)K);
    LAUNCH_OK();
    return 0;
}

int adkf_split_planes_t(const float* w, uint16_t* planes, int64_t K, int64_t N, void* stream) {
    (void)hipGetLastError();
    if (!w || !planes || K <= 0 || N <= 0 || (K & 1)) return ADKF_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(w) & 3) || (reinterpret_cast<uintptr_t>(planes) & 15) || ((N * K) & 7)) return ADKF_E_BADARG;
    if (K > 0x7fffffffLL || N > 0x7fffffffLL) return ADKF_E_SIZE;
    const size_t pairs = (size_t)(K / 2) * (size_t)N;
    if (pairs > (size_t)0x7fffffff * 256) return ADKF_E_SIZE;
    k_split3_t<<<(unsigned)((pairs + 255) / 256), 256, 0, static_cast<hipStream_t>(stream)>>>(w, planes, (int)K, (int)N);
    LAUNCH_OK();
    return 0;
}

// ADKF_DENSE_STREAM=0 (read once): k_dense3 for every K (A/B runs against k_dense3_sk; bit-identical results)
static bool dense_stream_forms() {
    static const bool on = [] { const char* e = getenv("ADKF_DENSE_STREAM"); return !(e && e[0] == '0'); }();
    return on;
}

int adkf_dense_forward(const float* x, int32_t ldx, const uint16_t* w_planes, const float* bias, float* y, int32_t ldy, int32_t M,
                       int32_t N, int32_t K, void* stream) {
    (void)hipGetLastError();
    if (!x || !w_planes || !y || M <= 0 || N <= 0 || K <= 0 || ldx < K || ldy < N) return ADKF_E_BADARG;
    if ((K % GK) || (ldx & 3)) return ADKF_E_SIZE;
    if ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(w_planes)) & 15) return ADKF_E_BADARG;
    static const bool optin = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense3), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  D3_LDS_BYTES) == hipSuccess;
    if (!optin) { g_last_hip_error = hipErrorInvalidValue; (void)hipGetLastError(); return ADKF_E_LAUNCH; }
    const long long tiles = (long long)ceil_div(M, D3_TM) * ceil_div(N, D3_TN);
    if (tiles > 0x7fffffffLL) return ADKF_E_SIZE;
    Dense3Args a{x, ldx, w_planes, (size_t)N * (size_t)K, bias, y, ldy, M, N, K};
    // short contractions over many rows: the persistent form that keeps a row tile's whole K extent in registers (bit-identical
    // results; 79 -> 63 us at 65 536 x 256 x 256, tools/x3_stream_bench.hip)
    const int tiles_m = ceil_div(M, D3_TM);
    if ((K == 64 || K == 128 || K == 256) && tiles_m >= num_cus() && dense_stream_forms()) {
        static const bool optin_sk = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense3_sk<2>), hipFuncAttributeMaxDynamicSharedMemorySize, D3_LDS_BYTES) == hipSuccess &&
                                     hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense3_sk<4>), hipFuncAttributeMaxDynamicSharedMemorySize, D3_LDS_BYTES) == hipSuccess &&
                                     hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense3_sk<8>), hipFuncAttributeMaxDynamicSharedMemorySize, D3_LDS_BYTES) == hipSuccess;
        if (optin_sk) {
            const unsigned grid = (unsigned)num_cus();
            hipStream_t st = static_cast<hipStream_t>(stream);
            if (K == 256) k_dense3_sk<8><<<grid, D3_NT, D3_LDS_BYTES, st>>>(a);
            else if (K == 128) k_dense3_sk<4><<<grid, D3_NT, D3_LDS_BYTES, st>>>(a);
            else k_dense3_sk<2><<<grid, D3_NT, D3_LDS_BYTES, st>>>(a);
            LAUNCH_OK();
            return 0;
        }
    }
    k_dense3<<<(unsigned)tiles, D3_NT, D3_LDS_BYTES, static_cast<hipStream_t>(stream)>>>(a);
    LAUNCH_OK();
    return 0;
}

// row ranges of the weight gradient: enough workgroups for ~4 rounds of the chip, ranges a multiple of the chunk, at most 64 of them
static int dense_tn_splits(int M, int N, int K, int* rows_per_split) {
    const long long tiles = (long long)ceil_div(N, D3_TM) * ceil_div(K, D3_TN);
    long long s = (4LL * num_cus() + tiles - 1) / tiles;
    const long long max_s = (M + 4 * GK - 1) / (4 * GK);          // at least four chunks per range
    if (s > max_s) s = max_s;
    if (s > 64) s = 64;
    if (s < 1) s = 1;
    int rps = (int)((M + s - 1) / s);
    rps = (rps + GK - 1) / GK * GK;
    *rows_per_split = rps;
    return ceil_div(M, rps);
}

size_t adkf_dense_weight_grad_scratch_bytes(int32_t M, int32_t N, int32_t K) {
    if (M <= 0 || N <= 0 || K <= 0) return 0;
    int rps;
    const int splits = dense_tn_splits(M, N, K, &rps);
    return sizeof(float) * (size_t)splits * (size_t)N * (size_t)K;
}

int adkf_dense_weight_grad(const float* g, int32_t ldg, const float* x, int32_t ldx, float* dw, int32_t M, int32_t N, int32_t K,
                           void* scratch, size_t scratch_bytes, void* stream) {
    (void)hipGetLastError();
    if (!g || !x || !dw || !scratch || M <= 0 || N <= 0 || K <= 0 || ldg < N || ldx < K) return ADKF_E_BADARG;
    if (scratch_bytes < adkf_dense_weight_grad_scratch_bytes(M, N, K)) return ADKF_E_WORKSPACE;
    static const bool optin = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense3_tn), hipFuncAttributeMaxDynamicSharedMemorySize,
                                                  D3_LDS_BYTES) == hipSuccess;
    if (!optin) { g_last_hip_error = hipErrorInvalidValue; (void)hipGetLastError(); return ADKF_E_LAUNCH; }
    int rps;
    const int splits = dense_tn_splits(M, N, K, &rps);
    const long long tiles = (long long)ceil_div(N, D3_TM) * ceil_div(K, D3_TN);
    if (tiles * splits > 0x7fffffffLL) return ADKF_E_SIZE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    Dense3TnArgs a{g, ldg, x, ldx, static_cast<float*>(scratch), M, N, K, rps};
    // ADKF_DENSE_TN_DEPTH (read once): 0 / unset = k_dense3_tn; 2 or 4 = k_dense3_tnd<2 | 4> (deeper prefetch, XCD-aware order; -2 / -4:
    // dispatch order).  Bit-identical partial sums.  Measured (tools/x3_stream_bench.hip, tools/r05_stream_prof.sh): alone on the chip
    // with its operands resident in the 256 MB MALL 73 -> 60 us at the C2 feature map, INSIDE the C2 step (operands from HBM) 66.9 / 68.4 /
    // 69.5 / 67.0 us for tn / tnd<4> / tnd<4> in dispatch order / tnd<2>: no gain where it is used, so the default stays k_dense3_tn.
    static const int tn_depth = [] { const char* e = getenv("ADKF_DENSE_TN_DEPTH"); return e ? atoi(e) : 0; }();
    const int depth = tn_depth < 0 ? -tn_depth : tn_depth;
    if (depth == 2 || depth == 4) {
        static const bool optin_d = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense3_tnd<2>), hipFuncAttributeMaxDynamicSharedMemorySize, D3_LDS_BYTES) == hipSuccess &&
                                    hipFuncSetAttribute(reinterpret_cast<const void*>(&k_dense3_tnd<4>), hipFuncAttributeMaxDynamicSharedMemorySize, D3_LDS_BYTES) == hipSuccess;
        if (!optin_d) { g_last_hip_error = hipErrorInvalidValue; (void)hipGetLastError(); return ADKF_E_LAUNCH; }
        const int sp = tn_depth < 0 ? -splits : splits;
        if (depth == 2) k_dense3_tnd<2><<<(unsigned)(tiles * splits), D3_NT, D3_LDS_BYTES, st>>>(a, (int)tiles, sp);
        else k_dense3_tnd<4><<<(unsigned)(tiles * splits), D3_NT, D3_LDS_BYTES, st>>>(a, (int)tiles, sp);
    } else k_dense3_tn<<<dim3((unsigned)tiles, (unsigned)splits), D3_NT, D3_LDS_BYTES, st>>>(a);
    const size_t n = (size_t)N * (size_t)K;
    k_dense3_reduce<<<(unsigned)((n + 255) / 256), 256, 0, st>>>(static_cast<const float*>(scratch), dw, n, splits);
    LAUNCH_OK();
    return 0;
}

int adkf_grad_sumsq(const float* g, int64_t n, float* partials, void* stream) {
    (void)hipGetLastError();
    if (!g || !partials || n <= 0 || (reinterpret_cast<uintptr_t>(g) & 15)) return ADKF_E_BADARG;
    k_grad_sumsq<<<SUMSQ_PARTS, STEP_NT, 0, static_cast<hipStream_t>(stream)>>>(g, (long)n, partials);
    LAUNCH_OK();
    return 0;
}

int adkf_clip_adam_step(float* p, float* g, float* m, float* v, int64_t n, const float* partials, int32_t n_partials, float scale,
                        float clip, double lr, double beta1, double beta2, double eps, double weight_decay, int32_t step, void* stream) {
    (void)hipGetLastError();
    if (!p || !g || !m || !v || !partials || n <= 0 || n_partials <= 0 || step <= 0) return ADKF_E_BADARG;
    if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15)
        return ADKF_E_BADARG;
    // bias corrections in double on the host, as torch.optim.Adam does for a python-number step
    const double bias1 = 1.0 - pow(beta1, (double)step);
    const double bias2_sqrt = sqrt(1.0 - pow(beta2, (double)step));
    AdamArgs a{p, g, m, v, (long)n, partials, n_partials, scale, clip, (float)(lr / bias1), (float)(1.0 - beta1), (float)beta2,
               (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)bias2_sqrt};
    const long n4 = (n + 3) / 4;
    int grid = (int)((n4 + STEP_NT - 1) / STEP_NT);
    grid = grid < 1 ? 1 : (grid > 2048 ? 2048 : grid);
    k_clip_adam<<<grid, STEP_NT, 0, static_cast<hipStream_t>(stream)>>>(a);
    LAUNCH_OK();
    return 0;
}

int adkf_clip_adam_step_one(float* p, float* g, float* m, float* v, int64_t n, float scale, float clip, double lr, double beta1, double beta2,
                            double eps, double weight_decay, int32_t step, uint16_t* planes_t, int32_t K, int32_t N, void* stream) {
    (void)hipGetLastError();
    if (!p || !g || !m || !v || n <= 0 || step <= 0) return ADKF_E_BADARG;
    if (n > CLIP_ADAM_ONE_MAX) return ADKF_E_SIZE;
    if ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) | reinterpret_cast<uintptr_t>(v)) & 15)
        return ADKF_E_BADARG;
    if (planes_t && (K <= 0 || N <= 0 || (int64_t)K * N != n || (K % STEP1_TILE) || (N % STEP1_TILE) || (reinterpret_cast<uintptr_t>(planes_t) & 3))) return ADKF_E_BADARG;
    const double bias1 = 1.0 - pow(beta1, (double)step);
    const double bias2_sqrt = sqrt(1.0 - pow(beta2, (double)step));
    AdamOneArgs o{{p, g, m, v, (long)n, nullptr, 0, scale, clip, (float)(lr / bias1), (float)(1.0 - beta1), (float)beta2,
                   (float)(1.0 - beta2), (float)eps, (float)weight_decay, (float)bias2_sqrt}, planes_t, K, N};
    const long n4 = (n + 3) / 4;
    int grid = (int)((n4 + STEP1_NT - 1) / STEP1_NT);
    if (grid < 1) grid = 1;
    if (planes_t) grid = (K / STEP1_TILE) * (N / STEP1_TILE);
    k_clip_adam_one<<<grid, STEP1_NT, 0, static_cast<hipStream_t>(stream)>>>(o);
    LAUNCH_OK();
    return 0;
}

int adkf_check_info(const int32_t* info, int32_t T, void* stream) {
    if (!info || T <= 0) return ADKF_E_BADARG;
    int32_t* host = static_cast<int32_t*>(malloc(sizeof(int32_t) * (size_t)T));
    if (!host) return ADKF_E_BADARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (hipMemcpyAsync(host, info, sizeof(int32_t) * (size_t)T, hipMemcpyDeviceToHost, st) != hipSuccess || hipStreamSynchronize(st) != hipSuccess) {
        free(host);
        return ADKF_E_LAUNCH;
    }
    int rc = 0;
    for (int t = 0; t < T; ++t) if (host[t] != 0) { rc = t + 1; break; }
    free(host);
    return rc;
}

}  // extern "C"
