// Problem functors for k_bgemm: every matrix product of the outer-NLL / Hessian / mixed-partial / dZ stages
// (oracle/closed_form.py names the same stages).  Operands that are elementwise functions of the squared
// distances (K_qs, dK/dl, Omega, A^-1 B_v ...) are generated while staging the tile, never stored.
//
// Every functor offers scalar accessors a(i,k) / b(k,j) and 4-wide ones a4 / b4 that fetch four consecutive
// entries along the operand's contiguous direction with ONE 16-byte load per source array (the tile loader uses
// them whenever TaskView::vec says all leading dimensions are multiples of 4 and the bases 16-byte aligned):
// the vector-memory pipe, not the matrix pipe, was the limiter with dword loads.
#pragma once
#include "gemm.h"

namespace adkf {

struct TaskView {
    // shared by all problems
    const int32_t* n_s;
    const int32_t* n_q;
    int ns_ld, nq_ld, vld, kind;
    const float* scal;  // [T, NSCAL]
    const float* vecs;  // [T, NVEC, vld]
    bool vec;           // 16-byte loads are legal on every array of this batch
    __device__ __forceinline__ int ns(int t) const { return n_s ? n_s[t] : ns_ld; }
    __device__ __forceinline__ int nq(int t) const { return n_q ? n_q[t] : nq_ld; }
    __device__ __forceinline__ const float* vec_ptr(int t, int which) const { return vecs + ((size_t)t * NVEC + which) * vld; }
};

__device__ __forceinline__ void ld4(const float* p, float (&v)[4]) {
    const float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}

__device__ __forceinline__ float4 ldq(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void unq(const float4& t, float (&v)[4]) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }

// ---- G1: P = Ainv * G,   G = dK_ss/dl = s kappa'(u) (-2u/l) ------------------------------------------
struct ProbP {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = true;
    static constexpr int NRED = 0;
    TaskView tv; const float* Ainv; const float* D2ss; float* P;
    int n; float os, ls, il2; const float *Ai, *D2; float* Po; bool vec;
    __device__ bool setup(int t) {
        n = tv.ns(t); const float* sc = tv.scal + (size_t)t * NSCAL; vec = tv.vec;
        os = sc[S_OS]; ls = sc[S_LS]; il2 = 1.f / (ls * ls);
        Ai = Ainv + (size_t)t * tv.ns_ld * tv.ns_ld; D2 = D2ss + (size_t)t * tv.ns_ld * tv.ns_ld; Po = P + (size_t)t * tv.ns_ld * tv.ns_ld;
        return n > 0;
    }
    __device__ int M() const { return n; } __device__ int N() const { return n; } __device__ int K() const { return n; }
    __device__ float gfun(float d2) const { float k0, k1, k2; const float u = d2 * il2; kappa3(tv.kind, u, k0, k1, k2); return os * k1 * u * (-2.f / ls); }
    __device__ float a(int i, int k) const { return Ai[(size_t)i * tv.ns_ld + k]; }
    __device__ float b(int k, int j) const { return gfun(D2[(size_t)j * tv.ns_ld + k]); }  // G symmetric: row j, contiguous in k
    static constexpr int A_NRAW = 1, B_NRAW = 1;   // two-phase operand path (gemm.h): loads, then arithmetic
    __device__ bool raw_ok() const { return true; }
    __device__ void a_raw(int i, int k, float4 (&r)[1]) const { r[0] = ldq(Ai + (size_t)i * tv.ns_ld + k); }
    __device__ void a_fin(int, int, const float4 (&r)[1], float (&v)[4]) const { unq(r[0], v); }
    __device__ void b_raw(int k, int j, float4 (&r)[1]) const { r[0] = ldq(D2 + (size_t)j * tv.ns_ld + k); }
    __device__ void b_fin(int, int, const float4 (&r)[1], float (&v)[4]) const {
        unq(r[0], v);
#pragma unroll
        for (int x = 0; x < 4; ++x) v[x] = gfun(v[x]);
    }
    __device__ void a4(int i, int k, float (&v)[4]) const { float4 r[1]; a_raw(i, k, r); a_fin(i, k, r, v); }
    __device__ void b4(int k, int j, float (&v)[4]) const { float4 r[1]; b_raw(k, j, r); b_fin(k, j, r, v); }
    __device__ void epi(int i, int j, float acc, float*) const { Po[(size_t)i * tv.ns_ld + j] = acc; }
    __device__ void store_red(int, const float*) const {}
};

// ---- G2: C = K_qs * Ainv ----------------------------------------------------------------------------
struct ProbC {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = true;
    static constexpr int NRED = 0;
    TaskView tv; const float* Ainv; const float* D2qs; float* C;
    int n, m; float os, il2; const float *Ai, *D2; float* Co; bool vec;
    __device__ bool setup(int t) {
        n = tv.ns(t); m = tv.nq(t); const float* sc = tv.scal + (size_t)t * NSCAL; vec = tv.vec;
        os = sc[S_OS]; il2 = 1.f / (sc[S_LS] * sc[S_LS]);
        Ai = Ainv + (size_t)t * tv.ns_ld * tv.ns_ld; D2 = D2qs + (size_t)t * tv.nq_ld * tv.ns_ld; Co = C + (size_t)t * tv.nq_ld * tv.ns_ld;
        return n > 0 && m > 0;
    }
    __device__ int M() const { return m; } __device__ int N() const { return n; } __device__ int K() const { return n; }
    __device__ float a(int i, int k) const { return os * kappa0(tv.kind, D2[(size_t)i * tv.ns_ld + k] * il2); }
    __device__ float b(int k, int j) const { return Ai[(size_t)j * tv.ns_ld + k]; }
    static constexpr int A_NRAW = 1, B_NRAW = 1;
    __device__ bool raw_ok() const { return true; }
    __device__ void a_raw(int i, int k, float4 (&r)[1]) const { r[0] = ldq(D2 + (size_t)i * tv.ns_ld + k); }
    __device__ void a_fin(int, int, const float4 (&r)[1], float (&v)[4]) const {
        unq(r[0], v);
#pragma unroll
        for (int x = 0; x < 4; ++x) v[x] = os * kappa0(tv.kind, v[x] * il2);
    }
    __device__ void b_raw(int k, int j, float4 (&r)[1]) const { r[0] = ldq(Ai + (size_t)j * tv.ns_ld + k); }
    __device__ void b_fin(int, int, const float4 (&r)[1], float (&v)[4]) const { unq(r[0], v); }
    __device__ void a4(int i, int k, float (&v)[4]) const { float4 r[1]; a_raw(i, k, r); a_fin(i, k, r, v); }
    __device__ void b4(int k, int j, float (&v)[4]) const { float4 r[1]; b_raw(k, j, r); b_fin(k, j, r, v); }
    __device__ void epi(int i, int j, float acc, float*) const { Co[(size_t)i * tv.ns_ld + j] = acc; }
    __device__ void store_red(int, const float*) const {}
};

// ---- G2b / G2c: ONE step of iterative refinement of C (working precision, the explicit inverse as the solver) -------
//      R = K_qs - C A            (ProbCres, A = s kappa(D2ss / l^2) + noise I generated on the fly)
//      C += R A^-1               (ProbCfix)
// The product with an explicit float32 inverse has an UNSTRUCTURED error of eps32 cond(A) |C|, which Sigma_q = K_qq - C K_sq
// turns into an absolute error on its small eigenvalues (seen: 2e-4 .. 6e-4 on dL/dZ for clustered low-dimensional features
// at noise 0.1, cond 150 .. 800).  One refinement step in working precision makes the solve backward stable (Skeel): C is
// then the exact solution for a slightly perturbed A and the Schur complement keeps its structure - what the reference's
// Cholesky solves give (tools/history/emulate_precision.py: configuration refC32+sweep has no failure below cond 1300; above, tasks
// take the float64 path of refine64.h).  Tasks whose (s + noise) max_i (A^-1)_ii stays below `thresh` skip both products
// (every benchmark configuration: 1.5 at C2).
struct ProbCres {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = true;
    static constexpr int NRED = 0;
    TaskView tv; const float* C; const float* D2ss; const float* D2qs; float* R; float thresh;
    int n, m; float os, il2, noise; const float *Ci, *Dss, *Dqs; float* Ro; bool vec;
    __device__ bool setup(int t) {
        n = tv.ns(t); m = tv.nq(t); const float* sc = tv.scal + (size_t)t * NSCAL; vec = tv.vec;
        if (tv.ns_ld <= 128 && !(sc[S_CONDA] > thresh)) return false;   // (beyond 128 points always: gated by S_CONDA, which large.h writes too, a 256-point task came out at 1.7e-4 on dL/dZ)
        os = sc[S_OS]; il2 = 1.f / (sc[S_LS] * sc[S_LS]); noise = sc[S_NOISE];
        Ci = C + (size_t)t * tv.nq_ld * tv.ns_ld; Dss = D2ss + (size_t)t * tv.ns_ld * tv.ns_ld;
        Dqs = D2qs + (size_t)t * tv.nq_ld * tv.ns_ld; Ro = R + (size_t)t * tv.nq_ld * tv.ns_ld;
        return n > 0 && m > 0;
    }
    __device__ int M() const { return m; } __device__ int N() const { return n; } __device__ int K() const { return n; }
    __device__ float afun(float d2, int k, int j) const { return os * kappa0(tv.kind, d2 * il2) + (k == j ? noise : 0.f); }
    __device__ float a(int i, int k) const { return Ci[(size_t)i * tv.ns_ld + k]; }
    __device__ float b(int k, int j) const { return afun(Dss[(size_t)j * tv.ns_ld + k], k, j); }   // A symmetric: row j, contiguous in k
    __device__ void a4(int i, int k, float (&v)[4]) const { ld4(Ci + (size_t)i * tv.ns_ld + k, v); }
    __device__ void b4(int k, int j, float (&v)[4]) const {
        ld4(Dss + (size_t)j * tv.ns_ld + k, v);
#pragma unroll
        for (int x = 0; x < 4; ++x) v[x] = afun(v[x], k + x, j);
    }
    __device__ void epi(int i, int j, float acc, float*) const {
        Ro[(size_t)i * tv.ns_ld + j] = os * kappa0(tv.kind, Dqs[(size_t)i * tv.ns_ld + j] * il2) - acc;
    }
    __device__ void store_red(int, const float*) const {}
};

struct ProbCfix {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = true;
    static constexpr int NRED = 0;
    TaskView tv; const float* R; const float* Ainv; float* C; float thresh;
    int n, m; const float *Ri, *Ai; float* Co; bool vec;
    __device__ bool setup(int t) {
        n = tv.ns(t); m = tv.nq(t); const float* sc = tv.scal + (size_t)t * NSCAL; vec = tv.vec;
        if (tv.ns_ld <= 128 && !(sc[S_CONDA] > thresh)) return false;   // (beyond 128 points always: gated by S_CONDA, which large.h writes too, a 256-point task came out at 1.7e-4 on dL/dZ)
        Ri = R + (size_t)t * tv.nq_ld * tv.ns_ld; Ai = Ainv + (size_t)t * tv.ns_ld * tv.ns_ld; Co = C + (size_t)t * tv.nq_ld * tv.ns_ld;
        return n > 0 && m > 0;
    }
    __device__ int M() const { return m; } __device__ int N() const { return n; } __device__ int K() const { return n; }
    __device__ float a(int i, int k) const { return Ri[(size_t)i * tv.ns_ld + k]; }
    __device__ float b(int k, int j) const { return Ai[(size_t)j * tv.ns_ld + k]; }
    __device__ void a4(int i, int k, float (&v)[4]) const { ld4(Ri + (size_t)i * tv.ns_ld + k, v); }
    __device__ void b4(int k, int j, float (&v)[4]) const { ld4(Ai + (size_t)j * tv.ns_ld + k, v); }
    __device__ void epi(int i, int j, float acc, float*) const { Co[(size_t)i * tv.ns_ld + j] += acc; }
    __device__ void store_red(int, const float*) const {}
};

// ---- G3: S = K_qq - C K_qs^T + noise I ----------------------------------------------------------------
struct ProbS {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = true;
    static constexpr int NRED = 0;
    TaskView tv; const float* C; const float* D2qs; const float* D2qq; float* S;
    int n, m; float os, il2, noise; const float *Ci, *Dqs, *Dqq; float* So; bool vec;
    __device__ bool setup(int t) {
        n = tv.ns(t); m = tv.nq(t); const float* sc = tv.scal + (size_t)t * NSCAL; vec = tv.vec;
        os = sc[S_OS]; il2 = 1.f / (sc[S_LS] * sc[S_LS]); noise = sc[S_NOISE];
        Ci = C + (size_t)t * tv.nq_ld * tv.ns_ld; Dqs = D2qs + (size_t)t * tv.nq_ld * tv.ns_ld;
        Dqq = D2qq + (size_t)t * tv.nq_ld * tv.nq_ld; So = S + (size_t)t * tv.nq_ld * tv.nq_ld;
        return n > 0 && m > 0;
    }
    __device__ int M() const { return m; } __device__ int N() const { return m; } __device__ int K() const { return n; }
    __device__ float a(int i, int k) const { return Ci[(size_t)i * tv.ns_ld + k]; }
    __device__ float b(int k, int j) const { return os * kappa0(tv.kind, Dqs[(size_t)j * tv.ns_ld + k] * il2); }
    static constexpr int A_NRAW = 1, B_NRAW = 1;
    __device__ bool raw_ok() const { return true; }
    __device__ void a_raw(int i, int k, float4 (&r)[1]) const { r[0] = ldq(Ci + (size_t)i * tv.ns_ld + k); }
    __device__ void a_fin(int, int, const float4 (&r)[1], float (&v)[4]) const { unq(r[0], v); }
    __device__ void b_raw(int k, int j, float4 (&r)[1]) const { r[0] = ldq(Dqs + (size_t)j * tv.ns_ld + k); }
    __device__ void b_fin(int, int, const float4 (&r)[1], float (&v)[4]) const {
        unq(r[0], v);
#pragma unroll
        for (int x = 0; x < 4; ++x) v[x] = os * kappa0(tv.kind, v[x] * il2);
    }
    __device__ void a4(int i, int k, float (&v)[4]) const { float4 r[1]; a_raw(i, k, r); a_fin(i, k, r, v); }
    __device__ void b4(int k, int j, float (&v)[4]) const { float4 r[1]; b_raw(k, j, r); b_fin(k, j, r, v); }
    // S is symmetric: tiles below the diagonal are written by their mirror images (exactly symmetric result)
    __device__ bool active(int m0, int n0) const { return m0 <= n0; }
    __device__ float value(int i, int j, float acc) const {
        return os * kappa0(tv.kind, Dqq[(size_t)i * tv.nq_ld + j] * il2) - acc + (i == j ? noise : 0.f);
    }
    __device__ void epi(int i, int j, float acc, float*) const {
        const float v = value(i, j, acc);
        So[(size_t)i * tv.nq_ld + j] = v;
        if ((i / GT) < (j / GT)) So[(size_t)j * tv.nq_ld + i] = v;
    }
    __device__ void epi4(int i0, int j, const float (&acc)[4], float* red) const {
        if (!(vec && (i0 / GT) < (j / GT))) {
#pragma unroll
            for (int r = 0; r < 4; ++r) epi(i0 + r, j, acc[r], red);
            return;
        }
        float v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { v[r] = value(i0 + r, j, acc[r]); So[(size_t)(i0 + r) * tv.nq_ld + j] = v[r]; }
        *reinterpret_cast<float4*>(So + (size_t)j * tv.nq_ld + i0) = make_float4(v[0], v[1], v[2], v[3]);
    }
    __device__ void store_red(int, const float*) const {}
};

// ---- G4: OC = Omega * C, Omega = (Sinv - e e^T)/2;  epilogue: W_qs and two reductions ----------------
struct ProbOC {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = false;
    static constexpr int NRED = 2;
    TaskView tv; const float* Sinv; const float* C; const float* D2qs; float* OC; float* Wqs; float* part; int ntiles; float dirscale;
    int n, m, t_; float os, ls, il2; const float *Si, *Ci, *Dqs, *ev, *al; float *OCo, *Wo; bool vec;
    __device__ bool setup(int t) {
        t_ = t; n = tv.ns(t); m = tv.nq(t); const float* sc = tv.scal + (size_t)t * NSCAL; vec = tv.vec;
        os = sc[S_OS]; ls = sc[S_LS]; il2 = 1.f / (ls * ls);
        Si = Sinv + (size_t)t * tv.nq_ld * tv.nq_ld; Ci = C + (size_t)t * tv.nq_ld * tv.ns_ld; Dqs = D2qs + (size_t)t * tv.nq_ld * tv.ns_ld;
        OCo = OC + (size_t)t * tv.nq_ld * tv.ns_ld; Wo = Wqs + (size_t)t * tv.nq_ld * tv.ns_ld;
        ev = tv.vec_ptr(t, V_E); al = tv.vec_ptr(t, V_ALPHA);
        return n > 0 && m > 0;
    }
    __device__ int M() const { return m; } __device__ int N() const { return n; } __device__ int K() const { return m; }
    __device__ float a(int i, int k) const { return 0.5f * (Si[(size_t)i * tv.nq_ld + k] - ev[i] * ev[k]); }
    __device__ float b(int k, int j) const { return Ci[(size_t)k * tv.ns_ld + j]; }
    static constexpr int A_NRAW = 3, B_NRAW = 1;
    __device__ bool raw_ok() const { return true; }
    __device__ void a_raw(int i, int k, float4 (&r)[3]) const {
        r[0] = ldq(Si + (size_t)i * tv.nq_ld + k); r[1] = ldq(ev + k); r[2].x = ev[i];
    }
    __device__ void a_fin(int, int, const float4 (&r)[3], float (&v)[4]) const {
        float e4[4];
        unq(r[0], v); unq(r[1], e4);
        const float ei = r[2].x;
#pragma unroll
        for (int x = 0; x < 4; ++x) v[x] = 0.5f * (v[x] - ei * e4[x]);
    }
    __device__ void b_raw(int k, int j, float4 (&r)[1]) const { r[0] = ldq(Ci + (size_t)k * tv.ns_ld + j); }
    __device__ void b_fin(int, int, const float4 (&r)[1], float (&v)[4]) const { unq(r[0], v); }
    __device__ void a4(int i, int k, float (&v)[4]) const { float4 r[3]; a_raw(i, k, r); a_fin(i, k, r, v); }
    __device__ void b4(int k, int j, float (&v)[4]) const { float4 r[1]; b_raw(k, j, r); b_fin(k, j, r, v); }
    __device__ void epi(int i, int j, float acc, float* red) const {
        OCo[(size_t)i * tv.ns_ld + j] = acc;
        const float MB = -2.f * acc - ev[i] * al[j];
        float k0, k1, k2; const float u = Dqs[(size_t)i * tv.ns_ld + j] * il2; kappa3(tv.kind, u, k0, k1, k2);
        Wo[(size_t)i * tv.ns_ld + j] = dirscale * MB * os * k1 * il2;
        red[0] += MB * k0;                          // -> d/ds  (sum M_B . kappa)
        red[1] += MB * os * k1 * u * (-2.f / ls);   // -> d/dl
    }
    __device__ void store_red(int tile, const float* red) const {
        part[((size_t)t_ * ntiles + tile) * 4 + 0] = red[0]; part[((size_t)t_ * ntiles + tile) * 4 + 1] = red[1];
    }
};

// ---- G5: M_A = C^T OC + sym(Cte alpha^T);  epilogue: W_ss (direct part) and three reductions ----------
struct ProbMA {
    static constexpr bool A_KCONTIG = false, B_KCONTIG = false;
    static constexpr int NRED = 3;
    TaskView tv; const float* C; const float* OC; const float* D2ss; float* Wss; float* part; int ntiles; float dirscale;
    int n, m, t_; float os, ls, il2; const float *Ci, *OCi, *Dss, *cte, *al; float* Wo; bool vec;
    __device__ bool setup(int t) {
        t_ = t; n = tv.ns(t); m = tv.nq(t); const float* sc = tv.scal + (size_t)t * NSCAL; vec = tv.vec;
        os = sc[S_OS]; ls = sc[S_LS]; il2 = 1.f / (ls * ls);
        Ci = C + (size_t)t * tv.nq_ld * tv.ns_ld; OCi = OC + (size_t)t * tv.nq_ld * tv.ns_ld; Dss = D2ss + (size_t)t * tv.ns_ld * tv.ns_ld;
        Wo = Wss + (size_t)t * tv.ns_ld * tv.ns_ld; cte = tv.vec_ptr(t, V_CTE); al = tv.vec_ptr(t, V_ALPHA);
        return n > 0 && m > 0;
    }
    __device__ int M() const { return n; } __device__ int N() const { return n; } __device__ int K() const { return m; }
    __device__ float a(int i, int k) const { return Ci[(size_t)k * tv.ns_ld + i]; }
    __device__ float b(int k, int j) const { return OCi[(size_t)k * tv.ns_ld + j]; }
    static constexpr int A_NRAW = 1, B_NRAW = 1;
    __device__ bool raw_ok() const { return true; }
    __device__ void a_raw(int i, int k, float4 (&r)[1]) const { r[0] = ldq(Ci + (size_t)k * tv.ns_ld + i); }
    __device__ void a_fin(int, int, const float4 (&r)[1], float (&v)[4]) const { unq(r[0], v); }
    __device__ void b_raw(int k, int j, float4 (&r)[1]) const { r[0] = ldq(OCi + (size_t)k * tv.ns_ld + j); }
    __device__ void b_fin(int, int, const float4 (&r)[1], float (&v)[4]) const { unq(r[0], v); }
    __device__ void a4(int i, int k, float (&v)[4]) const { float4 r[1]; a_raw(i, k, r); a_fin(i, k, r, v); }
    __device__ void b4(int k, int j, float (&v)[4]) const { float4 r[1]; b_raw(k, j, r); b_fin(k, j, r, v); }
    __device__ void epi(int i, int j, float acc, float* red) const {
        const float MA = acc + 0.5f * (cte[i] * al[j] + al[i] * cte[j]);
        float k0, k1, k2; const float u = Dss[(size_t)i * tv.ns_ld + j] * il2; kappa3(tv.kind, u, k0, k1, k2);
        Wo[(size_t)i * tv.ns_ld + j] = dirscale * MA * os * k1 * il2;
        if (i == j) red[0] += MA;
        red[1] += MA * k0;
        red[2] += MA * os * k1 * u * (-2.f / ls);
    }
    __device__ void store_red(int tile, const float* red) const {
        float* p = part + ((size_t)t_ * ntiles + tile) * 4;
        p[0] = red[0]; p[1] = red[1]; p[2] = red[2];
    }
};

// ---- G6: XA = (A^-1 B_v) A^-1;  epilogue: W_ss -= corr * (weights of the mixed partial term) ------------
struct ProbMixed {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = true;
    static constexpr int NRED = 0;
    TaskView tv; const float* Ainv; const float* P; const float* D2ss; float* Wss; float corrscale;
    int n; float os, ls, il2, noise, cn, cs, cl; const float *Ai, *Pi, *Dss, *al, *wv; float* Wo; bool vec;
    __device__ bool setup(int t) {
        n = tv.ns(t); const float* sc = tv.scal + (size_t)t * NSCAL; vec = tv.vec;
        os = sc[S_OS]; ls = sc[S_LS]; il2 = 1.f / (ls * ls); noise = sc[S_NOISE]; cn = sc[S_CN]; cs = sc[S_CS]; cl = sc[S_CL];
        Ai = Ainv + (size_t)t * tv.ns_ld * tv.ns_ld; Pi = P + (size_t)t * tv.ns_ld * tv.ns_ld; Dss = D2ss + (size_t)t * tv.ns_ld * tv.ns_ld;
        Wo = Wss + (size_t)t * tv.ns_ld * tv.ns_ld; al = tv.vec_ptr(t, V_ALPHA); wv = tv.vec_ptr(t, V_W);
        return n > 0;
    }
    __device__ int M() const { return n; } __device__ int N() const { return n; } __device__ int K() const { return n; }
    __device__ float a(int i, int k) const {
        const float ai = Ai[(size_t)i * tv.ns_ld + k];
        return (cn - cs * noise) * ai + (i == k ? cs : 0.f) + cl * Pi[(size_t)i * tv.ns_ld + k];
    }
    __device__ float b(int k, int j) const { return Ai[(size_t)j * tv.ns_ld + k]; }
    static constexpr int A_NRAW = 2, B_NRAW = 1;
    __device__ bool raw_ok() const { return true; }
    __device__ void a_raw(int i, int k, float4 (&r)[2]) const { r[0] = ldq(Ai + (size_t)i * tv.ns_ld + k); r[1] = ldq(Pi + (size_t)i * tv.ns_ld + k); }
    __device__ void a_fin(int i, int k, const float4 (&r)[2], float (&v)[4]) const {
        float p4[4];
        unq(r[0], v); unq(r[1], p4);
#pragma unroll
        for (int x = 0; x < 4; ++x) v[x] = (cn - cs * noise) * v[x] + (i == k + x ? cs : 0.f) + cl * p4[x];
    }
    __device__ void b_raw(int k, int j, float4 (&r)[1]) const { r[0] = ldq(Ai + (size_t)j * tv.ns_ld + k); }
    __device__ void b_fin(int, int, const float4 (&r)[1], float (&v)[4]) const { unq(r[0], v); }
    __device__ void a4(int i, int k, float (&v)[4]) const { float4 r[2]; a_raw(i, k, r); a_fin(i, k, r, v); }
    __device__ void b4(int k, int j, float (&v)[4]) const { float4 r[1]; b_raw(k, j, r); b_fin(k, j, r, v); }
    __device__ void epi(int i, int j, float acc, float*) const {
        const float fn = (float)n;
        const float dgdA = (-0.5f * acc + 0.5f * (wv[i] * al[j] + al[i] * wv[j])) / fn;
        const float Q = 0.5f * (Ai[(size_t)i * tv.ns_ld + j] - al[i] * al[j]) / fn;
        float k0, k1, k2; const float u = Dss[(size_t)i * tv.ns_ld + j] * il2; kappa3(tv.kind, u, k0, k1, k2);
        const float dBv = cs * os * k1 + cl * os * (-2.f / ls) * (k1 + u * k2);
        Wo[(size_t)i * tv.ns_ld + j] -= corrscale * (dgdA * os * k1 * il2 + Q * dBv * il2);
    }
    __device__ void store_red(int, const float*) const {}
};

// ---- G7: dZ = coef . Z - [W-weighted sums of Z]  (two K segments: support rows then query rows) --------
// dZs_i = coef_s[i] Zs_i - sum_k 4 Wss[i,k] Zs_k - sum_q 2 Wqs[q,i] Zq_q
// dZq_i = coef_q[i] Zq_i - sum_k 2 Wqs[i,k] Zs_k - sum_q 4 Wqq[i,q] Zq_q          (W_ss, W_qq symmetric)
template <bool QUERY>
struct ProbDZ {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = false;
    static constexpr int NRED = 0;
    TaskView tv; const float* Wss; const float* Wqs; const float* Wqq; const float* Zs; const float* Zq; float* dZ; int d;
    int n, m, rs_base; const float *Wssi, *Wqsi, *Wqqi, *Zsi, *Zqi, *rs; float* dZo; bool vec;
    __device__ bool setup(int t) {
        n = tv.ns(t); m = Wqs ? tv.nq(t) : 0; vec = tv.vec;
        Wssi = Wss ? Wss + (size_t)t * tv.ns_ld * tv.ns_ld : nullptr;
        Wqsi = Wqs ? Wqs + (size_t)t * tv.nq_ld * tv.ns_ld : nullptr;
        Wqqi = Wqq ? Wqq + (size_t)t * tv.nq_ld * tv.nq_ld : nullptr;
        Zsi = Zs + (size_t)t * tv.ns_ld * d; Zqi = Zq ? Zq + (size_t)t * tv.nq_ld * d : nullptr;
        dZo = dZ + (size_t)t * (QUERY ? tv.nq_ld : tv.ns_ld) * d;
        rs = nullptr; rs_base = 0;
        return QUERY ? (m > 0) : (n > 0);
    }
    __device__ int M() const { return QUERY ? m : n; } __device__ int N() const { return d; } __device__ int K() const { return n + m; }
    __device__ float a(int i, int k) const {
        if (!QUERY) return k < n ? 4.f * Wssi[(size_t)i * tv.ns_ld + k] : 2.f * Wqsi[(size_t)(k - n) * tv.ns_ld + i];
        return k < n ? 2.f * Wqsi[(size_t)i * tv.ns_ld + k] : 4.f * Wqqi[(size_t)i * tv.nq_ld + (k - n)];
    }
    __device__ float b(int k, int j) const { return k < n ? Zsi[(size_t)k * d + j] : Zqi[(size_t)(k - n) * d + j]; }
    __device__ void a4(int i, int k, float (&v)[4]) const {
        if (k + 3 < n) {
            ld4((QUERY ? Wqsi : Wssi) + (size_t)i * tv.ns_ld + k, v);
            const float f = QUERY ? 2.f : 4.f;
#pragma unroll
            for (int x = 0; x < 4; ++x) v[x] *= f;
        } else if (QUERY && k >= n && ((n & 3) == 0)) {
            ld4(Wqqi + (size_t)i * tv.nq_ld + (k - n), v);
#pragma unroll
            for (int x = 0; x < 4; ++x) v[x] *= 4.f;
        } else {
#pragma unroll
            for (int x = 0; x < 4; ++x) v[x] = a(i, k + x);  // W_qs^T segment (strided) or a group straddling the segments
        }
    }
    __device__ void b4(int k, int j, float (&v)[4]) const { ld4((k < n ? Zsi + (size_t)k * d : Zqi + (size_t)(k - n) * d) + j, v); }
    // two-phase path: the support count must be a multiple of the chunk, so that a chunk never straddles the two K segments
    static constexpr int A_NRAW = 1, B_NRAW = 1;
    __device__ bool raw_ok() const { return (n % GK) == 0; }
    __device__ void a_raw(int i, int k, float4 (&r)[1]) const {
        if (k < n) r[0] = ldq((QUERY ? Wqsi : Wssi) + (size_t)i * tv.ns_ld + k);
        else if (QUERY) r[0] = ldq(Wqqi + (size_t)i * tv.nq_ld + (k - n));
        else {   // W_qs^T: strided in k
            const float* w = Wqsi + (size_t)(k - n) * tv.ns_ld + i;
            r[0] = make_float4(w[0], w[tv.ns_ld], w[2 * (size_t)tv.ns_ld], w[3 * (size_t)tv.ns_ld]);
        }
    }
    __device__ void a_fin(int, int k, const float4 (&r)[1], float (&v)[4]) const {
        const float f = (k < n) == QUERY ? 2.f : 4.f;   // W_qs carries 2, W_ss and W_qq carry 4
        unq(r[0], v);
#pragma unroll
        for (int x = 0; x < 4; ++x) v[x] *= f;
    }
    __device__ void b_raw(int k, int j, float4 (&r)[1]) const { r[0] = ldq((k < n ? Zsi + (size_t)k * d : Zqi + (size_t)(k - n) * d) + j); }
    __device__ void b_fin(int, int, const float4 (&r)[1], float (&v)[4]) const { unq(r[0], v); }
    // coef_i = sum_k a(i, k): summed by the GEMM kernel itself while it stages this tile's A operand (gemm.h: set_rowsum) - the
    // separate row / column sum kernels over W_ss, W_qs, W_qq and their coefficient vectors are gone
    __device__ void set_rowsum(const float* a, int m0) { rs = a; rs_base = m0; }
    __device__ void epi(int i, int j, float acc, float*) const {
        const float z = QUERY ? Zqi[(size_t)i * d + j] : Zsi[(size_t)i * d + j];
        dZo[(size_t)i * d + j] = rs[i - rs_base] * z - acc;
    }
    __device__ void store_red(int, const float*) const {}
};

}  // namespace adkf
