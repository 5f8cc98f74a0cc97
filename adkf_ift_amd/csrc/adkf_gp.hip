// libadkf_gp.so - host side of the C ABI declared in include/adkf_gp.h: argument checks, workspace
// carving and the kernel pipeline of each entry point.  gfx950 only; no allocation, no synchronisation
// (except adkf_check_info), everything enqueued on the caller's stream.
#include <hip/hip_runtime.h>

#include "../../include/adkf_gp.h"
#include "large.h"

using namespace adkf;

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

template <class P>
void launch_gemm(const P& p, int T, int M, int N, hipStream_t st) {
    if (tile_edge(M, N) == GTL) {
        const int tm = ceil_div(M, GTL), tn = ceil_div(N, GTL);
        k_bgemm<P, GTL><<<((T + 7) / 8) * 8 * tm * tn, 256, 0, st>>>(p, T, tm, tn);
    } else {
        const int tm = ceil_div(M, GT), tn = ceil_div(N, GT);
        k_bgemm<P, GT><<<((T + 7) / 8) * 8 * tm * tn, 256, 0, st>>>(p, T, tm, tn);
    }
}

struct Workspace {
    float *mean, *nrm_s, *nrm_q, *D2ss, *D2qs, *D2qq, *Ainv, *P, *C, *S, *OC, *Wss, *Wqs, *Wqq, *vecs, *scal, *part_oc, *part_ma, *l0;
    // blocked path only (max(ns, nq) > REG_POINTS)
    float *lg_Dinv, *lg_C, *lg_F, *lg_logdet, *lg_part;
    int32_t *lg_info, *lg_med;  // lg_med: prefix[T], rank[T], hist[T, 256]
    FitShared* lg_fit;
    int vld, nt_oc, nt_ma;
    size_t bytes;
};

Workspace carve(void* base, int T, int ns, int nq, int d) {
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t nfloat) { float* p = base ? reinterpret_cast<float*>(static_cast<char*>(base) + off) : nullptr; off += align_up(nfloat * sizeof(float)); return p; };
    const size_t Tz = (size_t)T;
    w.vld = ns > nq ? ns : nq;
    w.nt_oc = nq > 0 ? tiles_of(nq, ns) : 0;   // per-tile partial reductions of ProbOC / ProbMA
    w.nt_ma = tiles_of(ns, ns);
    w.mean = take(Tz * d);
    w.nrm_s = take(Tz * ns);
    w.nrm_q = take(Tz * (nq > 0 ? nq : 1));
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
    w.lg_Dinv = w.lg_C = w.lg_F = w.lg_logdet = w.lg_part = nullptr; w.lg_info = w.lg_med = nullptr; w.lg_fit = nullptr;
    if (w.vld > REG_POINTS) {
        w.lg_Dinv = take(Tz * LB * LB);
        w.lg_C = take(Tz * LB * w.vld);
        w.lg_F = take(Tz * LB * w.vld);
        w.lg_logdet = take(Tz);
        const size_t tq = (size_t)((nq + GT - 1) / GT) * ((nq + GT - 1) / GT);
        const size_t ts = (size_t)((ns + GT - 1) / GT) * ((ns + GT - 1) / GT);
        w.lg_part = take(Tz * (ts > tq ? ts : tq) * 8);
        w.lg_info = reinterpret_cast<int32_t*>(take(Tz));
        w.lg_med = reinterpret_cast<int32_t*>(take(Tz * 258));
        w.lg_fit = reinterpret_cast<FitShared*>(take(Tz * ((sizeof(FitShared) + 3) / 4)));
    }
    w.bytes = off;
    return w;
}

int check_batch(const adkf_batch_t* b, bool need_query) {
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

#define LAUNCH_OK() do { if (hipGetLastError() != hipSuccess) return ADKF_E_LAUNCH; } while (0)

inline bool has_query(const adkf_batch_t* b) { return b->nq_max > 0 && b->Z_q != nullptr; }

// Stage A: centring, row norms, squared distances.  Skipped when the caller promises (ADKF_BATCH_REUSE_DIST) that
// this workspace already holds them for exactly this batch.
int stage_dist(const adkf_batch_t* b, const Workspace& w, bool with_query, hipStream_t st) {
    if (b->flags & ADKF_BATCH_REUSE_DIST) return 0;
    const int T = b->T, ns = b->ns_max, nq = with_query ? b->nq_max : 0, d = b->d;
    k_colmean<<<dim3(ceil_div(d, 64), T), 256, 0, st>>>(b->Z_s, b->n_s, ns, d, w.mean, T);
    k_rownorm<<<dim3(ceil_div(ns, 4), T), 256, 0, st>>>(b->Z_s, b->n_s, ns, d, w.mean, w.nrm_s, T);
    if (with_query) k_rownorm<<<dim3(ceil_div(nq, 4), T), 256, 0, st>>>(b->Z_q, b->n_q, nq, d, w.mean, w.nrm_q, T);
    ProbDist p;
    p.mean = w.mean; p.d = d; p.vec = vec_ok(b, w);
    {
        p.X = b->Z_s; p.Y = b->Z_s; p.nx = w.nrm_s; p.ny = w.nrm_s; p.n_x = b->n_s; p.n_y = b->n_s; p.x_ld = ns; p.y_ld = ns; p.symmetric = true; p.D2 = w.D2ss;
        launch_gemm(p, T, ns, ns, st);
    }
    if (with_query) {
        p.X = b->Z_q; p.Y = b->Z_s; p.nx = w.nrm_q; p.ny = w.nrm_s; p.n_x = b->n_q; p.n_y = b->n_s; p.x_ld = nq; p.y_ld = ns; p.symmetric = false; p.D2 = w.D2qs;
        launch_gemm(p, T, nq, ns, st);
        p.X = b->Z_q; p.Y = b->Z_q; p.nx = w.nrm_q; p.ny = w.nrm_q; p.n_x = b->n_q; p.n_y = b->n_q; p.x_ld = nq; p.y_ld = nq; p.symmetric = true; p.D2 = w.D2qq;
        launch_gemm(p, T, nq, nq, st);
    }
    LAUNCH_OK();
    return 0;
}

template <int NMAX, int NT>
void launch_inner_k(const InnerArgs& a, hipStream_t st) {
    if (a.kind == ADKF_KERNEL_RBF) k_inner<NMAX, NT, 0><<<grid_for(a.T, 1), NT, 0, st>>>(a);
    else k_inner<NMAX, NT, 1><<<grid_for(a.T, 1), NT, 0, st>>>(a);
}

LgMat lg_mat(const Workspace& w, float* M, int ld, const int32_t* n_arr, const FitShared* fit, int T) {
    LgMat m;
    m.M = M; m.ld = ld; m.n_arr = n_arr; m.fit = fit;
    m.Dinv = w.lg_Dinv; m.Cbuf = w.lg_C; m.Fbuf = w.lg_F; m.logdet = w.lg_logdet; m.info = w.lg_info;
    m.T = T; m.vec = (ld & 3) == 0;
    return m;
}

// M -> -(M^-1) in place by 128-pivot block steps (large.h)
void lg_sweep(const LgMat& m, hipStream_t st) {
    const int nb = ceil_div(m.ld, LB), tn = ceil_div(m.ld, GT);
    for (int step = 0; step < nb; ++step) {
        k_lg_diag<<<grid_for(m.T, 1), 512, 0, st>>>(m, step);
        ProbLgPanel pp; pp.m = m; pp.step = step;
        k_bgemm<ProbLgPanel><<<grid_for(m.T, 2 * tn), 256, 0, st>>>(pp, m.T, 2, tn);
        ProbLgUpdate pu; pu.m = m; pu.step = step;
        k_bgemm<ProbLgUpdate><<<grid_for(m.T, tn * tn), 256, 0, st>>>(pu, m.T, tn, tn);
    }
}

int launch_inner_large(const InnerArgs& a, const Workspace& w, hipStream_t st) {
    LgInner li;
    li.in = a; li.fit = w.lg_fit; li.part = w.lg_part;
    li.tiles_1d = ceil_div(a.ld, GT); li.ntiles = li.tiles_1d * li.tiles_1d;
    li.mat = lg_mat(w, a.Ainv, a.ld, a.n_s, w.lg_fit, a.T);
    LgMatvecArgs mv{li.mat, a.y_s, (size_t)a.ld, a.vecs + (size_t)V_ALPHA * a.vld, (size_t)NVEC * a.vld, -1.f};
    k_lg_begin<<<ceil_div(a.T, 64), 64, 0, st>>>(li);
    const int n_evals = a.max_evals > 0 ? a.max_evals : 1;
    for (int e = 0; e < n_evals; ++e) {
        k_lg_build<<<grid_for(a.T, li.ntiles), 256, 0, st>>>(li);
        lg_sweep(li.mat, st);
        k_lg_matvec<<<dim3(ceil_div(a.ld, 4), a.T), 256, 0, st>>>(mv);
        k_lg_traces<<<grid_for(a.T, li.ntiles), 256, 0, st>>>(li);
        k_lg_advance<<<a.T, 64, 0, st>>>(li);
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
        LgOuterFin fin{a, w.lg_logdet, w.lg_info};
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

void launch_rowsums(const RowsumArgs& ra, const Workspace& w, hipStream_t st) {
    const TaskView& tv = ra.tv;
    if (w.vld <= REG_POINTS) { k_rowsums<<<grid_for(ra.T, 1), SMALL_NT, 0, st>>>(ra); return; }
    if (ra.Wqs) {
        LgColsumArgs cs{ra.Wqs, tv.ns_ld, (size_t)tv.nq_ld * tv.ns_ld, tv.n_q, tv.nq_ld, tv.n_s, tv.ns_ld, nullptr, 0,
                        ra.vecs + (size_t)V_CS_QS * tv.vld, (size_t)NVEC * tv.vld};
        k_lg_colsum<<<dim3(ceil_div(tv.ns_ld, 64), ra.T), 1024, 0, st>>>(cs);
    }
    const int rows = tv.ns_ld > tv.nq_ld ? tv.ns_ld : tv.nq_ld;
    k_lg_rowsums<<<dim3(ceil_div(rows, 4), ra.T), 256, 0, st>>>(ra);
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
    if (b->flags & ADKF_BATCH_REUSE_INNER) {
        // A^-1, alpha and the per-task scalars of phi are already in the workspace (left by adkf_fit)
        hipMemsetAsync(info, 0, sizeof(int32_t) * (size_t)T, st);
    } else {
        InnerArgs ia = inner_args(b, w, const_cast<float*>(phi), info);
        rc = launch_inner(ia, w, st);
        if (rc) return rc;
    }
    TaskView tv = make_tv(b, w, true);
    const int tms = ceil_div(ns, GT), tmq = ceil_div(nq, GT);
    const float dirscale = (flags & ADKF_IGNORE_DIRECT_GRAD) ? 0.f : 1.f;
    const float corrscale = (with_hessian && !(flags & ADKF_IGNORE_GRAD_CORRECTION)) ? 1.f : 0.f;
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
    ProbC pc; pc.tv = tv; pc.Ainv = w.Ainv; pc.D2qs = w.D2qs; pc.C = w.C;
    launch_gemm(pc, T, nq, ns, st);
    ProbS ps; ps.tv = tv; ps.C = w.C; ps.D2qs = w.D2qs; ps.D2qq = w.D2qq; ps.S = w.S;
    launch_gemm(ps, T, nq, nq, st);
    OuterArgs oa{tv, w.C, w.S, b->y_s, b->y_q, w.vecs, w.scal, f_out, info, T};
    rc = launch_outer_factor(oa, w, nq, st);
    if (rc) return rc;
    ProbOC po; po.tv = tv; po.Sinv = w.S; po.C = w.C; po.D2qs = w.D2qs; po.OC = w.OC; po.Wqs = w.Wqs; po.part = w.part_oc; po.ntiles = w.nt_oc; po.dirscale = dirscale;
    launch_gemm(po, T, nq, ns, st);
    ProbMA pm; pm.tv = tv; pm.C = w.C; pm.OC = w.OC; pm.D2ss = w.D2ss; pm.Wss = w.Wss; pm.part = w.part_ma; pm.ntiles = w.nt_ma; pm.dirscale = dirscale;
    launch_gemm(pm, T, ns, ns, st);
    WqqArgs wq{tv, w.S, w.D2qq, w.Wqq, w.scal, dirscale, T};
    if (nq > REG_POINTS) {
        LgWqq lw{wq, w.lg_part, tmq * tmq, tmq};
        k_lg_wqq<<<grid_for(T, tmq * tmq), 256, 0, st>>>(lw);
        k_lg_wqq_fin<<<T, 64, 0, st>>>(lw);
    } else {
        k_wqq<<<grid_for(T, 1), SMALL_NT, 0, st>>>(wq);
    }
    SolveArgs sa{tv, w.scal, w.vecs, w.part_oc, w.part_ma, w.nt_oc, w.nt_ma, flags, g_phi_out, v_out, H_out, T, with_hessian ? 1 : 0};
    k_solve_v<<<T, 64, 0, st>>>(sa);
    if (corrscale != 0.f) {
        ProbMixed px; px.tv = tv; px.Ainv = w.Ainv; px.P = w.P; px.D2ss = w.D2ss; px.Wss = w.Wss; px.corrscale = corrscale;
        launch_gemm(px, T, ns, ns, st);
    }
    if (dZ_s || dZ_q) {
        RowsumArgs ra{tv, w.Wss, w.Wqs, w.Wqq, w.vecs, T};
        launch_rowsums(ra, w, st);
        if (dZ_s) {
            hipMemsetAsync(dZ_s, 0, (size_t)T * ns * d * sizeof(float), st);
            ProbDZ<false> pz; pz.tv = tv; pz.Wss = w.Wss; pz.Wqs = w.Wqs; pz.Wqq = w.Wqq; pz.Zs = b->Z_s; pz.Zq = b->Z_q; pz.dZ = dZ_s; pz.d = d;
            launch_gemm(pz, T, ns, d, st);
        }
        if (dZ_q) {
            hipMemsetAsync(dZ_q, 0, (size_t)T * nq * d * sizeof(float), st);
            ProbDZ<true> pz; pz.tv = tv; pz.Wss = w.Wss; pz.Wqs = w.Wqs; pz.Wqq = w.Wqq; pz.Zs = b->Z_s; pz.Zq = b->Z_q; pz.dZ = dZ_q; pz.d = d;
            launch_gemm(pz, T, nq, d, st);
        }
    }
    LAUNCH_OK();
    return 0;
}

}  // namespace

extern "C" {

const char* adkf_version(void) { return "adkf_gp 0.1 (gfx950)"; }

int adkf_max_points(void) { return MAX_POINTS; }

size_t adkf_workspace_bytes(int32_t T, int32_t ns_max, int32_t nq_max, int32_t d) {
    if (T <= 0 || ns_max <= 0 || nq_max < 0 || d <= 0) return 0;
    return carve(nullptr, T, ns_max, nq_max, d).bytes;
}

int adkf_median_lengthscale(const adkf_batch_t* b, float* l0, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_batch(b, false);
    if (rc) return rc;
    if (!l0 || !ws) return ADKF_E_BADARG;
    Workspace w = carve(ws, b->T, b->ns_max, b->nq_max, b->d);
    if (ws_bytes < w.bytes) return ADKF_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = stage_dist(b, w, has_query(b), st);
    if (rc) return rc;
    if (b->ns_max <= 128) k_median<512, 32><<<grid_for(b->T, 1), 512, 0, st>>>(w.D2ss, b->n_s, b->ns_max, l0, b->T);
    else if (b->ns_max <= 256) k_median<1024, 64><<<grid_for(b->T, 1), 1024, 0, st>>>(w.D2ss, b->n_s, b->ns_max, l0, b->T);
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

int adkf_init_params(const adkf_batch_t* b, int32_t use_numeric_labels, int32_t use_lengthscale_prior, float* phi,
                     float* priors, float* l0, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_batch(b, false);
    if (rc) return rc;
    if (!phi || !priors || !ws) return ADKF_E_BADARG;
    Workspace w = carve(ws, b->T, b->ns_max, b->nq_max, b->d);
    if (ws_bytes < w.bytes) return ADKF_E_WORKSPACE;
    float* l0p = l0 ? l0 : w.l0;
    rc = adkf_median_lengthscale(b, l0p, ws, ws_bytes, stream);
    if (rc) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    k_init_params<<<ceil_div(b->T, 64), 64, 0, st>>>(l0p, b->T, use_numeric_labels, use_lengthscale_prior, phi, priors);
    LAUNCH_OK();
    return 0;
}

int adkf_mll_value_grad(const adkf_batch_t* b, const float* phi, float* f_in, float* g_phi, float* dZ_s, int32_t* info,
                        void* ws, size_t ws_bytes, void* stream) {
    int rc = check_batch(b, false);
    if (rc) return rc;
    if (!phi || !f_in || !info || !ws || !b->y_s || !b->priors) return ADKF_E_BADARG;
    Workspace w = carve(ws, b->T, b->ns_max, b->nq_max, b->d);
    if (ws_bytes < w.bytes) return ADKF_E_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    rc = stage_dist(b, w, has_query(b), st);
    if (rc) return rc;
    InnerArgs ia = inner_args(b, w, const_cast<float*>(phi), info);
    ia.f_out = f_in; ia.g_out = g_phi;
    rc = launch_inner(ia, w, st);
    if (rc) return rc;
    if (dZ_s) {
        TaskView tv = make_tv(b, w, false);
        WinArgs wa{tv, w.Ainv, w.D2ss, w.Wss, w.scal, b->T};
        k_win<<<grid_for(b->T, 1), 256, 0, st>>>(wa);
        RowsumArgs ra{tv, w.Wss, nullptr, nullptr, w.vecs, b->T};
        launch_rowsums(ra, w, st);
        hipMemsetAsync(dZ_s, 0, (size_t)b->T * b->ns_max * b->d * sizeof(float), st);
        ProbDZ<false> pz; pz.tv = tv; pz.Wss = w.Wss; pz.Wqs = nullptr; pz.Wqq = nullptr; pz.Zs = b->Z_s; pz.Zq = nullptr; pz.dZ = dZ_s; pz.d = b->d;
        launch_gemm(pz, b->T, b->ns_max, b->d, st);
        LAUNCH_OK();
    }
    return 0;
}

int adkf_fit(const adkf_batch_t* b, float* phi, const adkf_fit_options_t* opt, float* f_final, float* gnorm,
             int32_t* n_evals, int32_t* info, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_batch(b, false);
    if (rc) return rc;
    if (!phi || !opt || !info || !ws || !b->y_s || !b->priors || opt->max_evals < 2) return ADKF_E_BADARG;
    Workspace w = carve(ws, b->T, b->ns_max, b->nq_max, b->d);
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
    return rc;
}

int adkf_predict(const adkf_batch_t* b, const float* phi, float* mean, float* var, float* cov, int32_t* info, void* ws,
                 size_t ws_bytes, void* stream) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    if (!phi || !mean || !info || !ws || !b->y_s || !b->priors) return ADKF_E_BADARG;
    Workspace w = carve(ws, b->T, b->ns_max, b->nq_max, b->d);
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
    TaskView tv = make_tv(b, w, true);
    const int T = b->T;
    ProbC pc; pc.tv = tv; pc.Ainv = w.Ainv; pc.D2qs = w.D2qs; pc.C = w.C;
    launch_gemm(pc, T, b->nq_max, b->ns_max, st);
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

int adkf_outer_nll_value_grad(const adkf_batch_t* b, const float* phi, float* f_out, float* g_phi, float* dZ_s,
                              float* dZ_q, int32_t* info, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    if (!phi || !f_out || !info || !ws || !b->y_s || !b->y_q || !b->priors) return ADKF_E_BADARG;
    Workspace w = carve(ws, b->T, b->ns_max, b->nq_max, b->d);
    if (ws_bytes < w.bytes) return ADKF_E_WORKSPACE;
    return outer_pipeline(b, w, phi, 0, false, f_out, dZ_s, dZ_q, g_phi, nullptr, nullptr, info, static_cast<hipStream_t>(stream));
}

int adkf_ift_hypergrad(const adkf_batch_t* b, const float* phi, int32_t flags, float* f_out, float* dZ_s, float* dZ_q,
                       float* g_phi_out, float* v, float* H, int32_t* info, void* ws, size_t ws_bytes, void* stream) {
    int rc = check_batch(b, true);
    if (rc) return rc;
    if (!phi || !f_out || !dZ_s || !dZ_q || !info || !ws || !b->y_s || !b->y_q || !b->priors) return ADKF_E_BADARG;
    Workspace w = carve(ws, b->T, b->ns_max, b->nq_max, b->d);
    if (ws_bytes < w.bytes) return ADKF_E_WORKSPACE;
    return outer_pipeline(b, w, phi, flags, true, f_out, dZ_s, dZ_q, g_phi_out, v, H, info, static_cast<hipStream_t>(stream));
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
