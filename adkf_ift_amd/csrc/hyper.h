// k_hyper: the whole outer / hypergradient stage of ONE task in ONE workgroup (round 4).
//
// Until round 3 the stage between the inner fit and the dL/dZ products was sixteen launches - k_alpha_refine, ProbP, k_hess, ProbC,
// ProbCres, ProbCfix, ProbS, k_outer_factor, k_refine64, ProbOC, ProbMA, k_wqq (+ k_solve_v), ProbMixed - each of which filled the
// chip for less than one round of workgroups: six N^3-sized batched GEMMs of four K-chunks per workgroup (prologue / epilogue
// bound, 19 - 31 us each at C2), three one-workgroup-per-task latency kernels (85 us) and five launches that test a device flag and
// leave (24 us): 260 us of a 1.05 ms step for 1.6 GFLOP.  Here a task's 512 lanes keep everything between HBM and the three weight
// matrices on chip:
//
//   C^T = A^-1 K_sq            (-> C row-major in LDS;  flagged tasks: one step of iterative refinement, ProbCres / ProbCfix)
//   S   = K_qq + noise I - C K_sq         in the accumulator layout of the matrix-pipe sweep (factor_m.h), swept in place
//   r = y_q - C y_s,  e = S^-1 r,  f_out,  C^T e;   W_qq and its three reductions straight from the sweep's registers
//   Omega C = (S^-1 C - e (C^T e)^T) / 2  -> W_qs, two reductions          (oracle/closed_form.py::outer_stage)
//   M_A = C^T (Omega C) + sym(C^T e alpha^T) -> W_ss (direct part, kept in registers), three reductions
//   with the Hessian:  beta, gamma, delta,  P = A^-1 G,  the nine traces -> H (3 x 3),  v = H^-1 grad_phi f_out, w
//                      (A^-1 B_v) A^-1 -> the mixed-partial part of W_ss    (::inner_stage, ::mixed_stage)
//
// i.e. fs_mol/models/adaptive_dkt.py:183-191 (f_outer) and fs_mol/utils/cauchy_hypergradient.py:120-161 for one task, six
// 128^3 products + one sweep per workgroup.  Every product runs on v_mfma_f32_16x16x4_f32 (exact fp32) with BOTH operands in LDS:
// wave w owns rows 16 w .. 16 w + 15 of the result as eight 16 x 16 accumulator tiles - the layout the sweep works in, so S never
// leaves the registers between its product and its factorisation.  Two operand regions of 128 x 144 floats:
//   * K-contiguous images ([row][k], ld = 132): fragments by 16-byte reads - the four k of a 16-byte group feed four consecutive
//     MFMAs (lane group g takes k = 16 kb + 4 g + j in MFMA j: any assignment of k to lane groups is a valid product as long as
//     both operands use the same one), conflict-free;
//   * an accumulator matrix goes back to LDS TRANSPOSED as eight 16-byte stores per lane (the lane's four registers of a tile are
//     four consecutive rows = four consecutive k of the transposed image): the formulation below is chosen such that every
//     product finds its operands either symmetric (A^-1, S^-1, G), loaded from HBM row-major (K_qs), or as the transposed image
//     of an earlier result (C from C^T, (Omega C)^T, P^T); where a result is needed along its rows as well (C: B operand of
//     S^-1 C, A operand of C^T (Omega C)) its image has ld = 144 (= 16 mod 32) so that the dword reads down a column are
//     conflict-free too.
// What an element-wise pass costs here: the CU retires 64 lane-instructions per cycle, a 128 x 128 matrix is 256 cycles PER INSTRUCTION
// per element, and a pass that loads its squared distance element by element, clamps the address and calls expf is ~70 of them:
// 18 k cycles, more than the product it follows (16.4 k: the matrix pipe's floor) - the first version of this kernel spent 52 of
// its 151 us there (tools/hyper_bench.hip).  Hence: (1) the exponential of a squared distance is taken ONCE per matrix and
// layout - K_qs while its image is loaded, K_qq in the epilogue that forms S, K_ss in the epilogue of M_A - and kept in
// registers (32 per matrix) for the later passes over the same matrix (W_qq; the traces and the mixed-partial weights; K_qs goes
// from its LDS image into registers before the image is overwritten); (2) the image of G = dK/dl is written from those registers
// (transposed 16-byte stores: G is symmetric) instead of being loaded and exponentiated again; (3) symmetric distance blocks are
// read as the transposed 16-byte groups, four rows of the accumulator layout at once; (4) full 128-point batches (FULL) use affine
// addresses without clamps; (5) the second copy of A^-1 is fetched into registers before the product in front of it.
// Sizes: support and query counts up to 128 (64 < max <= 128 takes this kernel; smaller batches keep the small-size kernels,
// larger ones the blocked path); 16-byte alignment as TaskView::vec says, else the element-wise loads below.
#pragma once
#include "kernels.h"

namespace adkf {

constexpr int HY_NT = 512, HY_N = 128;
constexpr int HY_LDK = 132;                 // K-contiguous images
constexpr int HY_LDM = 144;                 // images that are also read down their columns
constexpr int HY_BUF = HY_N * HY_LDM;       // floats per operand region
constexpr int HY_NVEC = 10;                 // alpha, y_s, y_q, e, cte, beta, gamma, delta, w, tmp
constexpr int HY_LDS_FLOATS = 2 * HY_BUF + HY_NVEC * HY_N + 9 * (HY_NT / 64) + 8;
constexpr size_t HY_LDS_BYTES = sizeof(float) * HY_LDS_FLOATS;

struct HyperArgs {
    TaskView tv;
    const float *Ainv, *D2ss, *D2qs, *D2qq, *y_s, *y_q, *priors;
    float *Wss, *Wqs, *Wqq;
    float *stash_ss, *stash_qs, *stash_qq;   // [T, 128, 128] each (FULL only): exponential factors parked between passes (workspace matrices P, OC, S: free here)
    float* vecs; float* scal; float* f_out; int32_t* info;
    float *g_phi_out, *v_out, *H_out;
    int T, reset_info, with_hessian, flags;
    float dirscale, corrscale, refine_thresh;
};

// ---- the kernel function through its exponential factor (the same expressions as kappa3, device_utils.h) -------------------
#ifndef ADKF_HY_FAST_EXP
#define ADKF_HY_FAST_EXP 0   // 1: ONE v_exp_f32 of a pre-scaled argument (what k_inner's search evaluations use, inner.h) instead of libm's expf.
                             // Experiment switch only (tools/r05_fastexp_ab.sh): REJECTED on parity - the golden Matern case gp_N128_Nq128_d256_k1_r1_s0
                             // goes from 1.9e-5 to 1.1e-4 on v (tolerance 1e-4): the outer stage needs the libm factor
#endif
#if ADKF_HY_FAST_EXP
// absolute error of the factor <= 2^-24 |arg| e^-|arg| <= 2.2e-8 (the rounding of the pre-scaled argument) + 1 ulp of the hardware exp2:
// below the float32 rounding of the matrix entries the factor goes into
__device__ __forceinline__ float hy_ex(int kind, float u) {
    return kind == 0 ? __builtin_amdgcn_exp2f(u * -0.72134752044448170368f) : __builtin_amdgcn_exp2f(__builtin_amdgcn_sqrtf(u) * -3.2259784787f);
}
#else
__device__ __forceinline__ float hy_ex(int kind, float u) { return kind == 0 ? expf(-0.5f * u) : expf(-SQRT5 * sqrtf(u)); }
#endif
__device__ __forceinline__ float hy_k0(int kind, float u, float ex) { return kind == 0 ? ex : (1.f + SQRT5 * sqrtf(u) + (5.f / 3.f) * u) * ex; }
__device__ __forceinline__ float hy_ex_of_k0(int kind, float u, float k0) { return kind == 0 ? k0 : k0 / (1.f + SQRT5 * sqrtf(u) + (5.f / 3.f) * u); }
__device__ __forceinline__ void hy_k3(int kind, float u, float ex, float& k0, float& k1, float& k2) {
    if (kind == 0) { k0 = ex; k1 = -0.5f * ex; k2 = 0.25f * ex; }
    else {
        const float r = sqrtf(u);
        k0 = (1.f + SQRT5 * r + (5.f / 3.f) * u) * ex;
        k1 = -(5.f / 6.f) * (1.f + SQRT5 * r) * ex;
        k2 = (25.f / 12.f) * ex;
    }
}

// ---- operand images -----------------------------------------------------------------------------------------------------
// buf[r][c] = f(src[r][c], r, c) for r < rows, c < cols, 0 elsewhere (r, c < 128).  Two halves so that the loads can be in flight
// across other work: hy_fetch issues all eight 16-byte loads of a lane (clamped addresses, no branch around a load), hy_put
// transforms and stores.  FULL (ld = 128, 16-byte alignment): no clamps, affine addresses.
// (Addresses: a uniform base plus a 32-bit lane offset that is made OPAQUE per call.  hipcc otherwise merges the identical address
// computations of the passes that visit the same matrix - A^-1 twice, D2_ss three times, the parked factors, W_ss - into 64-bit
// per-lane addresses that stay alive across the whole kernel, sixteen registers per matrix, and spills them.)
#define HY_OPAQUE1(v_) asm volatile("" : "+v"(v_))
// (the same for LDS: every helper derives its lane addresses from a copy of the thread index the optimiser cannot see through)
__device__ __forceinline__ int hy_tid() { int t = threadIdx.x; HY_OPAQUE1(t); return t; }
template <bool FULL>
__device__ __forceinline__ void hy_fetch(float4 (&v)[8], const float* src, int src_ld, int rows) {
    const int tid = threadIdx.x;
    unsigned base = (unsigned)((tid >> 5) * HY_N + (tid & 31) * 4);
    if (FULL) HY_OPAQUE1(base);
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int r = (tid >> 5) + 16 * u, c4 = (tid & 31) * 4;
        if (FULL) v[u] = *reinterpret_cast<const float4*>(src + (base + 16u * u * HY_N));
        else {
            const int rc = r < rows ? r : rows - 1, cc = c4 + 3 < src_ld ? c4 : 0;
            v[u] = *reinterpret_cast<const float4*>(src + (unsigned)(rc * src_ld + cc));   // (32-bit offset from the uniform base: no 64-bit address per load)
        }
    }
}
template <bool FULL, class F>
__device__ __forceinline__ void hy_put(float* buf, int ld, const float4 (&v)[8], int rows, int cols, F f) {
    const int tid = hy_tid();
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        const int r = (tid >> 5) + 16 * u, c4 = (tid & 31) * 4;
        float4 o;
        o.x = (FULL || (r < rows && c4 + 0 < cols)) ? f(v[u].x, r, c4 + 0) : 0.f;
        o.y = (FULL || (r < rows && c4 + 1 < cols)) ? f(v[u].y, r, c4 + 1) : 0.f;
        o.z = (FULL || (r < rows && c4 + 2 < cols)) ? f(v[u].z, r, c4 + 2) : 0.f;
        o.w = (FULL || (r < rows && c4 + 3 < cols)) ? f(v[u].w, r, c4 + 3) : 0.f;
        *reinterpret_cast<float4*>(buf + r * ld + c4) = o;
    }
}
template <bool FULL, class F>
__device__ __forceinline__ void hy_load(float* buf, int ld, const float* src, int src_ld, int rows, int cols, bool vec, F f) {
    if (FULL || vec) {
        float4 v[8];
        hy_fetch<FULL>(v, src, src_ld, rows);
        hy_put<FULL>(buf, ld, v, rows, cols, f);
    } else {
        for (int e = threadIdx.x; e < HY_N * HY_N; e += HY_NT) {
            const int r = e >> 7, c = e & 127;
            buf[r * ld + c] = (r < rows && c < cols) ? f(src[(unsigned)(r * src_ld + c)], r, c) : 0.f;
        }
    }
}

// A distance block in the ACCUMULATOR layout: dv[x][y] = D[16 w + 4 g + y][16 x + p] (clamped into rows x cols when not FULL).
// SYM: D is symmetric - FULL reads the transposed 16-byte group D[16 x + p][16 w + 4 g .. + 3] instead of four dwords.
template <bool FULL, bool SYM>
__device__ __forceinline__ void hy_dist(float (&dv)[8][4], const float* D, int ld, int rows, int cols) {
    const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4, w = threadIdx.x >> 6;
    if (FULL && SYM) {
        unsigned q = (unsigned)(p * HY_N + 16 * w + 4 * g);
        HY_OPAQUE1(q);
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const float4 t = *reinterpret_cast<const float4*>(D + (q + 16u * x * HY_N));
            dv[x][0] = t.x; dv[x][1] = t.y; dv[x][2] = t.z; dv[x][3] = t.w;
        }
    } else if (FULL) {
        unsigned q = (unsigned)((16 * w + 4 * g) * HY_N + p);
        HY_OPAQUE1(q);
#pragma unroll
        for (int x = 0; x < 8; ++x)
#pragma unroll
            for (int y = 0; y < 4; ++y) dv[x][y] = D[q + (unsigned)(y * HY_N + 16 * x)];
    } else {
        // clamped once per row and per column: four row offsets and eight column offsets, 32-bit, from the uniform base
        unsigned ro[4];
#pragma unroll
        for (int y = 0; y < 4; ++y) ro[y] = (unsigned)(min(16 * w + 4 * g + y, rows - 1) * ld);
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            const unsigned co = (unsigned)min(16 * x + p, cols - 1);
#pragma unroll
            for (int y = 0; y < 4; ++y) dv[x][y] = D[ro[y] + co];
        }
    }
}

// ---- the product ---------------------------------------------------------------------------------------------------------
// acc[x] += A[16 w .. +15][.] B[.][16 x .. +15] over k < 16 kblocks.  A_MN = false: A image is [row][k] (16-byte fragment reads);
// true: [k][row] (one dword per k).  B likewise ([col][k] / [k][col]).  The fragments of block kb + 1 are fetched before the 32
// MFMAs of block kb issue.
template <bool A_MN, bool B_MN>
struct HyFrag {
    float a[4];
    float b[8][4];
    __device__ __forceinline__ void fetch(const float* ap, int lda, const float* bp, int ldb, int kb) {
        if (A_MN) {
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] = ap[(16 * kb + j) * lda];
        } else {
            const float4 t = *reinterpret_cast<const float4*>(ap + 16 * kb);
            a[0] = t.x; a[1] = t.y; a[2] = t.z; a[3] = t.w;
        }
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            if (B_MN) {
#pragma unroll
                for (int j = 0; j < 4; ++j) b[x][j] = bp[(16 * kb + j) * ldb + 16 * x];
            } else {
                const float4 t = *reinterpret_cast<const float4*>(bp + 16 * x * ldb + 16 * kb);
                b[x][0] = t.x; b[x][1] = t.y; b[x][2] = t.z; b[x][3] = t.w;
            }
        }
    }
    __device__ __forceinline__ void multiply(f32x4_t (&acc)[8]) const {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int x = 0; x < 8; ++x) acc[x] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[x][j], acc[x], 0, 0, 0);
    }
};

// DB: the fragments of block kb + 1 are fetched before the 32 MFMAs of block kb issue (72 registers of fragments); DB = false
// (k_dz: four accumulator sets) fetches and multiplies in turn - the SIMD's other wave covers the LDS latency.
template <bool A_MN, bool B_MN, bool DB = true>
__device__ __forceinline__ void hy_gemm(f32x4_t (&acc)[8], const float* A, int lda, const float* B, int ldb, int kblocks) {
    const int tid = hy_tid(), lane = tid & 63, p = lane & 15, g = lane >> 4, w = tid >> 6;
    const float* ap = A_MN ? A + (4 * g) * lda + 16 * w + p : A + (16 * w + p) * lda + 4 * g;
    const float* bp = B_MN ? B + (4 * g) * ldb + p : B + p * ldb + 4 * g;
    if constexpr (!DB) {
#pragma unroll 1
        for (int kb = 0; kb < kblocks; ++kb) {
            HyFrag<A_MN, B_MN> f;
            f.fetch(ap, lda, bp, ldb, kb);
            f.multiply(acc);
        }
    } else {
        HyFrag<A_MN, B_MN> f0, f1;
        f0.fetch(ap, lda, bp, ldb, 0);
        int kb = 0;
#pragma unroll 1   // (fully unrolled - FULL has a constant trip count - hipcc hoists the fragment reads of all blocks and spills 390 registers)
        for (; kb + 2 <= kblocks; kb += 2) {
            f1.fetch(ap, lda, bp, ldb, kb + 1);
            f0.multiply(acc);
            if (kb + 2 < kblocks) f0.fetch(ap, lda, bp, ldb, kb + 2);
            f1.multiply(acc);
        }
        if (kb < kblocks) f0.multiply(acc);
    }
}

__device__ __forceinline__ void hy_zero(f32x4_t (&acc)[8]) {
#pragma unroll
    for (int x = 0; x < 8; ++x) acc[x] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
}

// buf[j][i] = sign * M[i][j] for the accumulator matrix M: eight 16-byte stores per lane (registers 0..3 of tile x = rows
// 16 w + 4 g + {0..3}, column 16 x + p)
__device__ __forceinline__ void hy_store_t(const f32x4_t (&acc)[8], float* buf, int ld, float sign = 1.f) {
    const int tid = hy_tid(), lane = tid & 63, p = lane & 15, g = lane >> 4, w = tid >> 6;
#pragma unroll
    for (int x = 0; x < 8; ++x)
        *reinterpret_cast<float4*>(buf + (16 * x + p) * ld + 16 * w + 4 * g) =
            make_float4(sign * acc[x][0], sign * acc[x][1], sign * acc[x][2], sign * acc[x][3]);
}

// A SYMMETRIC matrix in the accumulator layout to / from a row-major [128][128] global matrix as the transposed 16-byte groups
// (element (i, j) of the lane's tile x, registers 0..3, lands at G[16 x + p][16 w + 4 g + {0..3}] = its mirror image): 8 accesses per lane
__device__ __forceinline__ void hy_sym_store(float* G, const float (&v)[8][4]) {
    const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4, w = threadIdx.x >> 6;
    unsigned q = (unsigned)(p * HY_N + 16 * w + 4 * g);
    HY_OPAQUE1(q);
#pragma unroll
    for (int x = 0; x < 8; ++x) *reinterpret_cast<float4*>(G + (q + 16u * x * HY_N)) = make_float4(v[x][0], v[x][1], v[x][2], v[x][3]);
}
// ... one tile at a time, from inside a pass (the lane offset q as hy_sym_offset() gives it): what a tile leaves behind goes out
// at once instead of sitting in 32 registers until the pass is over
__device__ __forceinline__ unsigned hy_sym_offset() {
    const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4, w = threadIdx.x >> 6;
    unsigned q = (unsigned)(p * HY_N + 16 * w + 4 * g);
    HY_OPAQUE1(q);
    return q;
}
__device__ __forceinline__ void hy_sym_store_tile(float* G, unsigned q, int x, const float (&v)[4]) {
    *reinterpret_cast<float4*>(G + (q + 16u * x * HY_N)) = make_float4(v[0], v[1], v[2], v[3]);
}
__device__ __forceinline__ void hy_sym_load(float (&v)[8][4], const float* G) {
    const int lane = threadIdx.x & 63, p = lane & 15, g = lane >> 4, w = threadIdx.x >> 6;
    unsigned q = (unsigned)(p * HY_N + 16 * w + 4 * g);
    HY_OPAQUE1(q);
#pragma unroll
    for (int x = 0; x < 8; ++x) {
        const float4 t = *reinterpret_cast<const float4*>(G + (q + 16u * x * HY_N));
        v[x][0] = t.x; v[x][1] = t.y; v[x][2] = t.z; v[x][3] = t.w;
    }
}

// Sums K per-lane values over the workgroup: float32 inside a row of sixteen lanes (DPP), then the 32 row totals of each value in
// FLOAT64 by one thread per value (the old pipeline summed float32 per 64 x 64 tile and the tiles in float64; a float32 tree over
// all 16 384 elements of a matrix cost grad_phi f_out - three sums that nearly cancel - a factor two in accuracy on clustered
// tasks).  `part`: K * 32 floats, `out`: K doubles, visible to everybody on return (two barriers).
template <int K>
__device__ __forceinline__ void hy_sum(const float (&v)[K], float* part, double* out) {
    const int tid = hy_tid(), lane = tid & 63, w = tid >> 6;
#pragma unroll
    for (int q = 0; q < K; ++q) {
        float s = v[q];
        s += dpp_f<DPP_XOR1>(s); s += dpp_f<DPP_XOR2>(s); s += dpp_f<DPP_HALF_MIRROR>(s); s += dpp_f<DPP_MIRROR>(s);
        if ((lane & 15) == 0) part[q * 32 + w * 4 + (lane >> 4)] = s;
    }
    __syncthreads();
    if (tid < K) {
        double s = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const float4 t = *reinterpret_cast<const float4*>(part + tid * 32 + 4 * i);
            s += ((double)t.x + (double)t.y) + ((double)t.z + (double)t.w);
        }
        out[tid] = s;
    }
    __syncthreads();
}

// out[j] = sum_{i < rows} M[i][j] x[i] for the image M (any ld): 128 columns x 4 row parts, partials through `scratch` (512 floats;
// TWO: a second image / output with the same x, 1024 floats).  One barrier inside, a second at the end (out visible on return).
template <bool TWO>
__device__ __forceinline__ void hy_colsum(const float* M, int ld, const float* M2, int ld2, int rows, const float* x, float* scratch, float* out, float* out2) {
    const int tid = hy_tid(), j = tid & 127, part = tid >> 7;
    const int per = (rows + 3) >> 2, i0 = part * per, i1 = min(rows, i0 + per);
    float s0 = 0.f, s1 = 0.f, t0 = 0.f, t1 = 0.f;
    int i = i0;
    for (; i + 2 <= i1; i += 2) {
        const float x0 = x[i], x1 = x[i + 1];
        s0 = fmaf(M[i * ld + j], x0, s0); s1 = fmaf(M[(i + 1) * ld + j], x1, s1);
        if (TWO) { t0 = fmaf(M2[i * ld2 + j], x0, t0); t1 = fmaf(M2[(i + 1) * ld2 + j], x1, t1); }
    }
    if (i < i1) { s0 = fmaf(M[i * ld + j], x[i], s0); if (TWO) t0 = fmaf(M2[i * ld2 + j], x[i], t0); }
    scratch[tid] = s0 + s1;
    if (TWO) scratch[512 + tid] = t0 + t1;
    __syncthreads();
    if (tid < 128) out[tid] = (scratch[tid] + scratch[128 + tid]) + (scratch[256 + tid] + scratch[384 + tid]);
    else if (TWO && tid < 256) out2[tid - 128] = (scratch[512 + tid - 128] + scratch[512 + tid]) + (scratch[512 + 128 + tid] + scratch[512 + 256 + tid]);
    __syncthreads();
}

// Between the tiles of an element-wise pass: the passes are fully unrolled over a lane's 32 elements (the tiles are registers), and
// without a fence hipcc interleaves all 32 exponentials - some 400 registers of temporaries, i.e. 1.8 KB of scratch per lane
#define HY_FENCE() __builtin_amdgcn_sched_barrier(0)
// ... and the fence alone is not enough: the SLP vectoriser has merged the element chains of all tiles before the scheduler sees
// them.  An opaque re-definition of what a tile leaves behind (its reduction partials, its outputs) cuts the chains per tile;
// inside a tile the four elements still pair up into packed FP32 instructions.
#define HY_OPAQUE4(v_) asm volatile("" : "+v"((v_)[0]), "+v"((v_)[1]), "+v"((v_)[2]), "+v"((v_)[3]))
#define HY_OPAQUE3(a_, b_, c_) asm volatile("" : "+v"(a_), "+v"(b_), "+v"(c_))
#define HY_OPAQUE2(a_, b_) asm volatile("" : "+v"(a_), "+v"(b_))

template <bool FULL, int KIND>   // KIND: 0 RBF, 1 Matern-5/2 (compile time: with a run-time kind hipcc evaluates BOTH kernel functions for every element and selects)
__global__ __launch_bounds__(HY_NT, 1) void k_hyper(HyperArgs a) {
    using SW = Sweep<128, 512>;
    __shared__ SweepSmem<128, 512> sm;
    extern __shared__ __align__(16) float hy_lds[];
    int t, tile;
    if (!task_tile(a.T, 1, t, tile)) return;
    const int tid = threadIdx.x, lane = tid & 63, p = lane & 15, g = lane >> 4, w = tid >> 6;
    const int n = FULL ? HY_N : a.tv.ns(t), m = FULL ? HY_N : a.tv.nq(t), lds = FULL ? HY_N : a.tv.ns_ld, ldq = FULL ? HY_N : a.tv.nq_ld;
    const int vld = a.tv.vld;
    if (n <= 0 || m <= 0) return;
    float* X = hy_lds;
    float* Y = X + HY_BUF;
    float* al = Y + HY_BUF;      // alpha
    float* ysv = al + HY_N;      // y_s
    float* rv = ysv + HY_N;      // y_q (r lives in sm.vec_in)
    float* ev = rv + HY_N;       // e
    float* cte = ev + HY_N;      // C^T e
    float* be = cte + HY_N;      // beta, gamma, delta, w
    float* ga = be + HY_N;
    float* de = ga + HY_N;
    float* wv = de + HY_N;
    float* tmp = wv + HY_N;
    float* red = tmp + HY_N;     // 9 * 8 floats (block_sum) + 8 broadcast slots
    float* bc = red + 9 * (HY_NT / 64);
    double* dsum = reinterpret_cast<double*>(red);   // 9 doubles (hy_sum), the same 72 floats
    float* scratch = sm.scratch();   // 1024 floats, free between sweeps

    float* sc = a.scal + (size_t)t * NSCAL;
    const float noise = sc[S_NOISE], os = sc[S_OS], ls = sc[S_LS], il2 = 1.f / (ls * ls), gl = -2.f / ls;
    constexpr int kind = KIND;
    const bool vec = FULL || a.tv.vec;
    const float* Ai = a.Ainv + (size_t)t * lds * lds;
    const float* Dss = a.D2ss + (size_t)t * lds * lds;
    const float* Dqs = a.D2qs + (size_t)t * ldq * lds;
    const float* Dqq = a.D2qq + (size_t)t * ldq * ldq;
    float* st_ss = a.stash_ss + (size_t)t * HY_N * HY_N;   // (FULL only)
    float* st_qs = a.stash_qs + (size_t)t * HY_N * HY_N;
    float* st_qq = a.stash_qq + (size_t)t * HY_N * HY_N;
    float* vb = a.vecs + (size_t)t * NVEC * vld;
    const int kn = FULL ? 8 : (n + 15) >> 4, km = FULL ? 8 : (m + 15) >> 4;       // 16-wide k blocks that hold a real row
    const bool refine = sc[S_CONDA] > a.refine_thresh;       // (s + noise) max diag(A^-1): C and alpha get one refinement step
    const bool refine_alpha = refine && sc[S_AREF] == 0.f;
    // element (x, y) of this lane in the accumulator layout: row i = i0 + y, column j = j0 + 16 x
    const int i0 = 16 * w + 4 * g, j0 = p;
    auto in_s = [&](int i) { return FULL || i < n; };
    auto in_q = [&](int i) { return FULL || i < m; };

    if (tid < HY_N) {
        al[tid] = tid < n ? vb[V_ALPHA * vld + tid] : 0.f;
        ysv[tid] = tid < n ? a.y_s[(size_t)t * lds + tid] : 0.f;
        rv[tid] = tid < m ? a.y_q[(size_t)t * ldq + tid] : 0.f;
    }
    auto ident = [](float v, int, int) { return v; };
    auto k0fun = [=](float d2, int, int) { const float u = d2 * il2; return os * hy_k0(kind, u, hy_ex(kind, u)); };
    auto afun = [=](float d2, int r, int c) { const float u = d2 * il2; return os * hy_k0(kind, u, hy_ex(kind, u)) + (r == c ? noise : 0.f); };

    // ================================================ outer stage ==========================================================
    ADKF_SST(0);
    {
        float4 va[8], vk[8];
        hy_fetch<FULL>(va, Ai, lds, n);
        hy_fetch<FULL>(vk, Dqs, lds, m);
        if (FULL || vec) {
            hy_put<FULL>(Y, HY_LDK, va, n, n, ident);              // A^-1 (symmetric)
            hy_put<FULL>(X, HY_LDK, vk, m, n, k0fun);              // kappa(D2_qs / l^2) [m][n]: K_qs without its scale s
        }
        if (FULL) {   // ... and row-major into the workspace: the epilogue of Omega C reads it back in the accumulator layout
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = (tid >> 5) + 16 * u, c4 = (tid & 31) * 4;
                *reinterpret_cast<float4*>(st_qs + (unsigned)(r * HY_N + c4)) = *reinterpret_cast<const float4*>(X + r * HY_LDK + c4);
            }
        }
    }
    if (!(FULL || vec)) {
        hy_load<false>(Y, HY_LDK, Ai, lds, n, n, false, ident);
        hy_load<false>(X, HY_LDK, Dqs, lds, m, n, false, k0fun);
    }
    __syncthreads();
    ADKF_SST(1);
    f32x4_t acc[8];
    hy_zero(acc);
    hy_gemm<false, false>(acc, Y, HY_LDK, X, HY_LDK, kn);     // (C^T / s)[i][j] = sum_k A^-1[i][k] kappa_qs[j][k]
    ADKF_SST(2);
    if (refine) {   // uniform over the workgroup; none at C2
        // R^T = K_sq - A C^T,  C^T += A^-1 R^T   (ProbCres / ProbCfix; alpha likewise: k_alpha_refine)
        __syncthreads();
        hy_store_t(acc, Y, HY_LDM);                            // C row-major
        hy_load<FULL>(X, HY_LDK, Dss, lds, n, n, vec, afun);   // A
        __syncthreads();
        if (refine_alpha) {
            hy_colsum<false>(X, HY_LDK, nullptr, 0, n, al, scratch, tmp, nullptr);   // A alpha (A symmetric)
            if (tid < HY_N) tmp[tid] = tid < n ? ysv[tid] - tmp[tid] : 0.f;
        }
        f32x4_t rr[8];
        hy_zero(rr);
        hy_gemm<false, false>(rr, X, HY_LDK, Y, HY_LDM, kn);   // (A C^T)[i][j] = sum_k A[i][k] C[j][k]
#pragma unroll
        for (int x = 0; x < 8; ++x) {
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                const int i = i0 + y, j = j0 + 16 * x;
                const float u = Dqs[(unsigned)(min(j, m - 1) * lds + min(i, n - 1))] * il2;
                rr[x][y] = (i < n && j < m) ? os * hy_k0(kind, u, hy_ex(kind, u)) - rr[x][y] : 0.f;
            }
            HY_OPAQUE4(rr[x]);
            HY_FENCE();
        }
        __syncthreads();
        hy_store_t(rr, Y, HY_LDK);                             // R row-major
        hy_load<FULL>(X, HY_LDK, Ai, lds, n, n, vec, ident);
        __syncthreads();
        if (refine_alpha) {
            hy_colsum<false>(X, HY_LDK, nullptr, 0, n, tmp, scratch, be, nullptr);   // A^-1 (y - A alpha)
            if (tid < n) { al[tid] += be[tid]; vb[V_ALPHA * vld + tid] = al[tid]; }
            if (tid == 0) sc[S_AREF] = 1.f;
        }
        hy_gemm<false, false>(acc, X, HY_LDK, Y, HY_LDK, kn);  // C^T += A^-1 R^T
        __syncthreads();
        hy_load<FULL>(X, HY_LDK, Dqs, lds, m, n, vec, k0fun);
    }
    __syncthreads();
    hy_store_t(acc, Y, HY_LDM);                                // C row-major [m][n], ld 144
    __syncthreads();

    // r = y_q - C y_s: thread (row, quarter), the quarter's 16-byte groups interleaved so that the eight lanes of a read differ in bank
    {
        const int i = tid >> 2, q = tid & 3;
        float s = 0.f;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = 16 * u + 4 * q;
            const float4 c4 = *reinterpret_cast<const float4*>(Y + i * HY_LDM + k);
            const float4 y4 = *reinterpret_cast<const float4*>(ysv + k);
            s = fmaf(c4.x, y4.x, fmaf(c4.y, y4.y, fmaf(c4.z, y4.z, fmaf(c4.w, y4.w, s))));
        }
        s += dpp_f<DPP_XOR1>(s);
        s += dpp_f<DPP_XOR2>(s);
        if (q == 0) {
            sm.vec_in[i] = i < m ? rv[i] - s : 0.f;
            if (i < m) vb[V_MU * vld + i] = s;
        }
    }
    ADKF_SST(3);
    hy_zero(acc);
    hy_gemm<false, false>(acc, Y, HY_LDM, X, HY_LDK, kn);      // (C K_sq / s)[i][j] = sum_k C[i][k] kappa_qs[j][k]
    ADKF_SST(4);
    float mm[8][4];
    {
        float dv[8][4];
        hy_dist<FULL, true>(dv, Dqq, ldq, m, m);
        const unsigned sq = hy_sym_offset();
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            float eqq[4];
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                const int i = i0 + y, j = j0 + 16 * x;
                const float u = dv[x][y] * il2;
                eqq[y] = hy_ex(kind, u);
                const float sv = os * hy_k0(kind, u, eqq[y]) + (i == j ? noise : 0.f) - acc[x][y];
                mm[x][y] = (in_q(i) && in_q(j)) ? sv : (i == j ? 1.f : 0.f);
            }
            if (FULL) hy_sym_store_tile(st_qq, sq, x, eqq);    // the exponential factor of K_qq: parked for the W_qq pass behind the sweep
            HY_OPAQUE4(mm[x]);
            HY_FENCE();
        }
    }
    __syncthreads();                                           // vec_in complete; everybody is done with X and Y
    ADKF_SST(5);
    SW::run(mm, m, sm);                                        // mm = -(S^-1)
    ADKF_SST(6);
    float logdet;
    const int info = SW::finish(m, sm, logdet);
    const float pivr = pivot_ratio<HY_NT>(sm.pivs, m, sm.red);
    SW::solve(mm, sm.vec_in, sm.vec_out);                      // e = S^-1 r
    float qv[1] = {0.f};
    if (tid < HY_N) {
        const float e = tid < m ? sm.vec_out[tid] : 0.f, r = tid < m ? sm.vec_in[tid] : 0.f;
        ev[tid] = e;
        if (tid < m) { vb[V_E * vld + tid] = e; vb[V_R * vld + tid] = r; }
        qv[0] = r * e;
    }
    block_sum<1, HY_NT>(qv, red);                              // (barriers inside: ev is visible below)
    const float fval = 0.5f * qv[0] + 0.5f * logdet + 0.5f * (float)m * LOG_2PI;
    if (tid == 0) {
        sc[S_PIVR_S] = pivr; sc[S_FOUT] = fval; sc[S_LOGDETS] = logdet;
        if (a.f_out) a.f_out[t] = (info == 0) ? fval : NAN;
        if (a.reset_info) a.info[t] = info != 0 ? 100000 + info : 0;
        else if (info != 0 && a.info[t] == 0) a.info[t] = 100000 + info;
    }
    ADKF_SST(7);
    // W_qq and its reductions from the registers; S^-1 into X (the K_qs image is spent) for the next product
    float r8[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // qq0 qq1 qq2 | oc0 oc1 | ma0 ma1 ma2 | (spare)
    // a vector's entries at this lane's rows / columns of the accumulator layout; fetched from LDS where a pass starts, not kept in
    // registers across the products in between (12 registers per vector: with four of them resident the Matern and the ragged
    // instances spilled)
    auto rows_cols = [&](const float* vsrc, float (&vi)[4], float (&vj)[8]) {
#pragma unroll
        for (int y = 0; y < 4; ++y) vi[y] = vsrc[i0 + y];
#pragma unroll
        for (int x = 0; x < 8; ++x) vj[x] = vsrc[j0 + 16 * x];
    };
    {
        float e_i[4], e_j[8];
        rows_cols(ev, e_i, e_j);
        f32x4_t si[8];
#pragma unroll
        for (int x = 0; x < 8; ++x) si[x] = (f32x4_t){mm[x][0], mm[x][1], mm[x][2], mm[x][3]};
        hy_store_t(si, X, HY_LDK, -1.f);                       // S^-1 (identity beyond m)
        float* Wo = a.Wqq + (size_t)t * ldq * ldq;
        float dv[8][4], eqq[8][4];
        hy_dist<FULL, true>(dv, Dqq, ldq, m, m);
        if (FULL) hy_sym_load(eqq, st_qq);
        const float wsc = a.dirscale * os * il2;
        const unsigned sq = hy_sym_offset();
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            float wq[4];
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                const int i = i0 + y, j = j0 + 16 * x;
                wq[y] = 0.f;
                if (in_q(i) && in_q(j)) {
                    const float om = 0.5f * (-mm[x][y] - e_i[y] * e_j[x]);
                    float k0, k1, k2; const float u = dv[x][y] * il2; hy_k3(kind, u, FULL ? eqq[x][y] : hy_ex(kind, u), k0, k1, k2);
                    wq[y] = wsc * om * k1;
                    if (!FULL) Wo[(unsigned)(i * ldq + j)] = wq[y];
                    if (i == j) r8[0] += om;
                    r8[1] += om * k0;
                    r8[2] += om * os * k1 * u * gl;
                }
            }
            if (FULL) hy_sym_store_tile(Wo, sq, x, wq);        // (W_qq is symmetric: the transposed 16-byte groups)
            HY_OPAQUE3(r8[0], r8[1], r8[2]);
            HY_FENCE();
        }
    }
    ADKF_SST(8);
    hy_colsum<false>(Y, HY_LDM, nullptr, 0, m, ev, scratch, cte, nullptr);   // C^T e   (barriers inside: X is complete after them)
    if (tid < n) vb[V_CTE * vld + tid] = cte[tid];
    ADKF_SST(9);
    hy_zero(acc);
    hy_gemm<false, true>(acc, X, HY_LDK, Y, HY_LDM, km);       // (S^-1 C)[i][j] = sum_k S^-1[i][k] C[k][j]
    ADKF_SST(10);
    {
        float* Wo = a.Wqs + (size_t)t * ldq * lds;
        float e_i[4], e_j[8], al_i[4], al_j[8], cte_i[4], cte_j[8];
        rows_cols(ev, e_i, e_j);
        rows_cols(al, al_i, al_j);
        rows_cols(cte, cte_i, cte_j);
        float dv[8][4], kq[8][4];
        hy_dist<FULL, false>(dv, Dqs, lds, m, n);
        if (FULL) hy_dist<true, false>(kq, st_qs, HY_N, HY_N, HY_N);   // kappa_qs as parked by the load pass
        const float wsc = a.dirscale * os * il2;
#pragma unroll
        for (int x = 0; x < 8; ++x) {
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                const int i = i0 + y, j = j0 + 16 * x;
                float oc = 0.f;
                if (in_q(i) && in_s(j)) {
                    oc = 0.5f * (acc[x][y] - e_i[y] * cte_j[x]);
                    const float MB = -2.f * oc - e_i[y] * al_j[x];
                    float k0, k1, k2; const float u = dv[x][y] * il2;
                    hy_k3(kind, u, FULL ? hy_ex_of_k0(kind, u, kq[x][y] / os) : hy_ex(kind, u), k0, k1, k2);
                    Wo[(unsigned)(i * lds + j)] = wsc * MB * k1;
                    r8[3] += MB * k0;
                    r8[4] += MB * os * k1 * u * gl;
                }
                acc[x][y] = oc;
            }
            HY_OPAQUE2(r8[3], r8[4]); HY_OPAQUE4(acc[x]);
            HY_FENCE();
        }
    }
    ADKF_SST(11);
    __syncthreads();                                           // everybody has read S^-1
    hy_store_t(acc, X, HY_LDK);                                // (Omega C)^T row-major
    __syncthreads();
    hy_zero(acc);
    hy_gemm<true, false>(acc, Y, HY_LDM, X, HY_LDK, km);       // M_A[i][j] = sum_k C[k][i] (Omega C)[k][j]
    ADKF_SST(12);
    float* Wss_o = a.Wss + (size_t)t * lds * lds;
    {
        float al_i[4], al_j[8], cte_i[4], cte_j[8];
        rows_cols(al, al_i, al_j);
        rows_cols(cte, cte_i, cte_j);
        float dv[8][4];
        hy_dist<FULL, true>(dv, Dss, lds, n, n);
        const float wsc = a.dirscale * os * il2;
        const unsigned sq = hy_sym_offset();
#pragma unroll
        for (int x = 0; x < 8; ++x) {
            float ess[4], wss[4];
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                const int i = i0 + y, j = j0 + 16 * x;
                wss[y] = 0.f; ess[y] = 0.f;
                float gv = 0.f;
                if (in_s(i) && in_s(j)) {
                    const float MA = acc[x][y] + 0.5f * (cte_i[y] * al_j[x] + al_i[y] * cte_j[x]);
                    const float u = dv[x][y] * il2;
                    ess[y] = hy_ex(kind, u);
                    float k0, k1, k2; hy_k3(kind, u, ess[y], k0, k1, k2);
                    wss[y] = wsc * MA * k1;
                    if (!FULL) Wss_o[(unsigned)(i * lds + j)] = wss[y];
                    gv = os * k1 * u * gl;                      // G = dK/dl
                    if (i == j) r8[5] += MA;
                    r8[6] += MA * k0;
                    r8[7] += MA * gv;
                }
                acc[x][y] = gv;
            }
            // the direct part of W_ss goes out now (symmetric: transposed 16-byte groups); the mixed-partial pass updates it in place.
            // The exponential factor of K_ss is parked for the traces and that pass.
            if (FULL) { hy_sym_store_tile(Wss_o, sq, x, wss); hy_sym_store_tile(st_ss, sq, x, ess); }
            HY_OPAQUE3(r8[5], r8[6], r8[7]); HY_OPAQUE4(acc[x]);
            HY_FENCE();
        }
    }
    ADKF_SST(13);
    hy_sum<9>(r8, scratch, dsum);                              // (barriers inside: everybody is done with X and Y)
    // grad_phi f_out (solve_v_task): the three pieces of each component nearly cancel - float64 from the row totals on
    const float d1n = sc[S_D1N], d1s = sc[S_D1S], d1l = sc[S_D1L];
    const float g0 = (float)(dsum[0] + dsum[5]) * d1n;
    const float g1 = (float)(dsum[6] + dsum[3] + dsum[1]) * d1s;
    const float g2 = (float)(dsum[7] + dsum[4] + dsum[2]) * d1l;
    if (tid == 0) {
        sc[S_QQ_TR] = (float)dsum[0]; sc[S_QQ_K] = (float)dsum[1]; sc[S_QQ_L] = (float)dsum[2];
        sc[S_GOUT0] = g0; sc[S_GOUT1] = g1; sc[S_GOUT2] = g2;
        if (a.g_phi_out) { a.g_phi_out[t * 3 + 0] = g0; a.g_phi_out[t * 3 + 1] = g1; a.g_phi_out[t * 3 + 2] = g2; }
    }
    __syncthreads();                                           // (dsum is read; the Hessian stage reuses it)
    ADKF_SST(14);

    // ================================================ Hessian, v, mixed partial ==============================================
    float cn = 0.f, cs = 0.f, cl = 0.f;
    if (a.with_hessian) {
        hy_load<FULL>(Y, HY_LDK, Ai, lds, n, n, vec, ident);
        hy_store_t(acc, X, HY_LDK);                            // G (symmetric): from the registers of the epilogue above
        __syncthreads();
        ADKF_SST(15);
        hy_colsum<true>(X, HY_LDK, Y, HY_LDK, n, al, scratch, be, ga);        // beta = G alpha, gamma = A^-1 alpha
        hy_colsum<false>(Y, HY_LDK, nullptr, 0, n, be, scratch, de, nullptr);   // delta = A^-1 beta
        if (tid < n) { vb[V_BETA * vld + tid] = be[tid]; vb[V_GAMMA * vld + tid] = ga[tid]; vb[V_DELTA * vld + tid] = de[tid]; }
        ADKF_SST(16);
        hy_zero(acc);
        hy_gemm<false, false>(acc, Y, HY_LDK, X, HY_LDK, kn);  // P[i][j] = sum_k A^-1[i][k] G[j][k]
        ADKF_SST(17);
        float h9[9] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};   // trA2, trPA, trPP, trAinvKll, aKlla, ag, bg, bd, ab
        {
            float al_i[4], al_j[8];
            rows_cols(al, al_i, al_j);
            float dv[8][4], ess[8][4];
            hy_dist<FULL, true>(dv, Dss, lds, n, n);
            if (FULL) hy_sym_load(ess, st_ss);
#pragma unroll
            for (int x = 0; x < 8; ++x) {
#pragma unroll
                for (int y = 0; y < 4; ++y) {
                    const int i = i0 + y, j = j0 + 16 * x;
                    if (in_s(i) && in_s(j)) {
                        const float ai = Y[i * HY_LDK + j];
                        float k0, k1, k2; const float u = dv[x][y] * il2; hy_k3(kind, u, FULL ? ess[x][y] : hy_ex(kind, u), k0, k1, k2);
                        const float Kll = os * (k2 * 4.f * u * u + k1 * 6.f * u) * il2;
                        h9[0] += ai * ai; h9[1] += acc[x][y] * ai; h9[3] += ai * Kll; h9[4] += al_i[y] * al_j[x] * Kll;
                    }
                }
                HY_OPAQUE2(h9[0], h9[1]); HY_OPAQUE2(h9[3], h9[4]);
                HY_FENCE();
            }
        }
        ADKF_SST(18);
        __syncthreads();                                       // everybody has read G
        hy_store_t(acc, X, HY_LDK);                            // X[j][i] = P[i][j]: P^T row-major
        __syncthreads();
#pragma unroll
        for (int x = 0; x < 8; ++x) {
#pragma unroll
            for (int y = 0; y < 4; ++y) {
                const int i = i0 + y, j = j0 + 16 * x;
                if (in_s(i) && in_s(j)) h9[2] += acc[x][y] * X[i * HY_LDK + j];   // P_ij P_ji
            }
            asm volatile("" : "+v"(h9[2]));
            HY_FENCE();
        }
        if (tid < n) { h9[5] = al[tid] * ga[tid]; h9[6] = be[tid] * ga[tid]; h9[7] = be[tid] * de[tid]; h9[8] = al[tid] * be[tid]; }
        ADKF_SST(19);
        hy_sum<9>(h9, scratch, dsum);
        if (tid == 0) {
#pragma unroll
            for (int q = 0; q < 9; ++q) h9[q] = (float)dsum[q];
            hess_assemble(sc, a.priors + t * 4, n, h9);
            float v[3] = {0.f, 0.f, 0.f};
            if (!(a.flags & 1)) {
                // 3 x 3 elimination with partial pivoting in float64 (the reference: torch.linalg.solve, cauchy_hypergradient.py:136)
                double Mx[3][4];
                const float gg[3] = {g0, g1, g2};
                for (int i = 0; i < 3; ++i) { for (int j = 0; j < 3; ++j) Mx[i][j] = sc[S_H0 + i * 3 + j]; Mx[i][3] = gg[i]; }
                for (int c = 0; c < 3; ++c) {
                    int pv = c;
                    for (int r = c + 1; r < 3; ++r) if (fabs(Mx[r][c]) > fabs(Mx[pv][c])) pv = r;
                    if (pv != c) for (int j = 0; j < 4; ++j) { const double tm = Mx[c][j]; Mx[c][j] = Mx[pv][j]; Mx[pv][j] = tm; }
                    const double ip = 1.0 / Mx[c][c];
                    for (int r = c + 1; r < 3; ++r) { const double f = Mx[r][c] * ip; for (int j = c; j < 4; ++j) Mx[r][j] -= f * Mx[c][j]; }
                }
                double vd[3] = {0.0, 0.0, 0.0};
                for (int c = 2; c >= 0; --c) { double s = Mx[c][3]; for (int j = c + 1; j < 3; ++j) s -= Mx[c][j] * vd[j]; vd[c] = s / Mx[c][c]; }
                for (int c = 0; c < 3; ++c) v[c] = (float)vd[c];
            }
            sc[S_V0] = v[0]; sc[S_V1] = v[1]; sc[S_V2] = v[2];
            const float cn_ = v[0] * d1n, cs_ = v[1] * d1s / os, cl_ = v[2] * d1l;
            sc[S_CN] = cn_; sc[S_CS] = cs_; sc[S_CL] = cl_;
            bc[0] = cn_; bc[1] = cs_; bc[2] = cl_;
            if (a.v_out) { a.v_out[t * 3 + 0] = v[0]; a.v_out[t * 3 + 1] = v[1]; a.v_out[t * 3 + 2] = v[2]; }
            if (a.H_out) for (int q = 0; q < 9; ++q) a.H_out[t * 9 + q] = sc[S_H0 + q];
        }
        __syncthreads();
        cn = bc[0]; cs = bc[1]; cl = bc[2];
        if (tid < HY_N) {
            const float wval = tid < n ? cn * ga[tid] + cs * (al[tid] - noise * ga[tid]) + cl * de[tid] : 0.f;
            wv[tid] = wval;
            if (tid < n) vb[V_W * vld + tid] = wval;
        }
        ADKF_SST(20);
        if (a.corrscale != 0.f) {
            // X[k][i] <- B'[i][k] = (cn - cs noise) A^-1[i][k] + cs [i == k] + cl P[i][k]   (X holds P^T; A^-1 is symmetric)
            const float ca = cn - cs * noise;
            const int tb = hy_tid();
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int r = (tb >> 5) + 16 * u, c4 = (tb & 31) * 4;
                float4 xv = *reinterpret_cast<float4*>(X + r * HY_LDK + c4);
                const float4 yv = *reinterpret_cast<const float4*>(Y + r * HY_LDK + c4);
                xv.x = cl * xv.x + ca * yv.x + ((r == c4 + 0 && r < n) ? cs : 0.f);
                xv.y = cl * xv.y + ca * yv.y + ((r == c4 + 1 && r < n) ? cs : 0.f);
                xv.z = cl * xv.z + ca * yv.z + ((r == c4 + 2 && r < n) ? cs : 0.f);
                xv.w = cl * xv.w + ca * yv.w + ((r == c4 + 3 && r < n) ? cs : 0.f);
                *reinterpret_cast<float4*>(X + r * HY_LDK + c4) = xv;
            }
            __syncthreads();                                   // (w is visible too)
            ADKF_SST(21);
            hy_zero(acc);
            hy_gemm<true, false>(acc, X, HY_LDK, Y, HY_LDK, kn);   // ((A^-1 B_v) A^-1)[i][j] = sum_k B'[i][k] A^-1[j][k]
            ADKF_SST(22);
            float dv[8][4], ess[8][4], wss[8][4];
            hy_dist<FULL, true>(dv, Dss, lds, n, n);
            if (FULL) { hy_sym_load(ess, st_ss); hy_sym_load(wss, Wss_o); }
            const float ifn = 1.f / (float)n;
            const unsigned sq = hy_sym_offset();
            float w_i[4], w_j[8], al_i[4], al_j[8];
            rows_cols(wv, w_i, w_j);
            rows_cols(al, al_i, al_j);
#pragma unroll
            for (int x = 0; x < 8; ++x) {
#pragma unroll
                for (int y = 0; y < 4; ++y) {
                    const int i = i0 + y, j = j0 + 16 * x;
                    if (in_s(i) && in_s(j)) {
                        const float dgdA = (-0.5f * acc[x][y] + 0.5f * (w_i[y] * al_j[x] + al_i[y] * w_j[x])) * ifn;
                        const float Q = 0.5f * (Y[i * HY_LDK + j] - al_i[y] * al_j[x]) * ifn;
                        float k0, k1, k2; const float u = dv[x][y] * il2; hy_k3(kind, u, FULL ? ess[x][y] : hy_ex(kind, u), k0, k1, k2);
                        const float dBv = cs * os * k1 + cl * os * gl * (k1 + u * k2);
                        const float corr = a.corrscale * (dgdA * os * k1 * il2 + Q * dBv * il2);
                        if (FULL) wss[x][y] -= corr;
                        else Wss_o[(unsigned)(i * lds + j)] -= corr;
                    }
                }
                if (FULL) hy_sym_store_tile(Wss_o, sq, x, wss[x]);
                HY_FENCE();
            }
        }
    } else if (tid == 0) {
        sc[S_V0] = sc[S_V1] = sc[S_V2] = 0.f;
        sc[S_CN] = sc[S_CS] = sc[S_CL] = 0.f;
        if (a.v_out) { a.v_out[t * 3 + 0] = 0.f; a.v_out[t * 3 + 1] = 0.f; a.v_out[t * 3 + 2] = 0.f; }
        if (a.H_out) for (int q = 0; q < 9; ++q) a.H_out[t * 9 + q] = 0.f;
    }
    ADKF_SST(23);
    ADKF_SST(24);
}

}  // namespace adkf
