// k_dense3: C[M, N] = A[M, K] B[N, K]^T (+ bias) - a dense layer in torch's F.linear layout - with FP32 products on the BF16 matrix pipe
// (gemm_x3.h: three-way split operands, six v_mfma_f32_16x16x32_bf16 per 16 x 16 x 32 block, the leading term in an accumulator of its
// own), as a PIPELINED kernel: what gemm_x3.h section "what would take it further" describes.
//
//   * B (the weights: small, shared by every row tile) arrives PRE-SPLIT - three planes [N][K] of bfloat16 written once per weight
//     update by k_split3 - so staging it is three 16-byte loads and three 16-byte LDS stores per lane and chunk, no arithmetic;
//   * A (the activations) is split on the fly, eight consecutive k per lane from two 16-byte loads, one 16-byte LDS store per plane;
//   * 128 x 128 output tile, 512 lanes = 2 x 4 waves of 64 x 32 (accumulators 2 x 32 registers), K chunks of 32, LDS DOUBLE-buffered:
//     one barrier per chunk; the loads of chunk c + 2 are issued in iteration c and consumed in iteration c + 1;
//   * the two waves that share a SIMD run the halves of an iteration in OPPOSITE order (waves 0 - 3: stage chunk c + 1, then the MFMAs
//     of chunk c; waves 4 - 7: MFMAs first), so that one's splitting arithmetic and LDS traffic sit under the other's matrix instructions.
// Measured and not kept: a PERSISTENT form (one workgroup per CU walking its tiles with one pipeline over all their chunks, the next
// tile's first chunks staged under the last ones of the current tile): 73.5 against 73.7 us at 65 536 x 256 x 256, 2.65 against 2.50 - 2.57 ms
// at 56 554 x 3 072 x 1 408 - the short-K shape is bound by its 134 MB of HBM traffic plus the products, not by the tile boundaries.
// What bounds it (tools/x3_gemm_bench.hip with -DD3_ABLATE builds, 56 554 x 3 072 x 1 408: 2.50 ms): without the MFMAs 1.86 ms, without the
// operand loads after the first chunk 2.38, without the stores 2.42, with none of the three 0.73 - the operand traffic out of L2 (40 KB per
// tile and chunk: 18.7 GB per call, ~10 TB/s while nothing else runs; 3.7 GB of it from beyond L2, rocprofv3 FETCH_SIZE) takes as long as
// the products, and the two only partly overlap.  A second register set (loads three chunks ahead) changed nothing (2.54 ms); larger
// tiles would (256 x 128 needs a swizzled 64-byte row stride to fit the double buffer into 160 KB) - not built.
#pragma once
#include "gemm_x3.h"

namespace adkf {

#ifndef D3_PINGPONG
#define D3_PINGPONG 1
#endif
#ifndef D3_ABLATE   // diagnostics (tools/x3_gemm_bench.hip): 1 no operand loads after the first chunk, 2 no stores of the result, 4 no MFMAs
#define D3_ABLATE 0
#endif
#ifndef D3_EAGER_A
#define D3_EAGER_A 0
#endif
constexpr int D3_TM = 128, D3_TN = 128, D3_NT = 512, D3_PLANE = D3_TM * X3_RS;
constexpr int D3_LDS_BYTES = 2 * 2 * 3 * D3_PLANE * (int)sizeof(unsigned short);   // two buffers x (A, B) x three planes: 122 880

// x [rows, K] float -> three planes [3][rows][K] of bfloat16 (K a multiple of 2)
__global__ void k_split3(const float* __restrict__ x, unsigned short* __restrict__ planes, size_t n_pairs, size_t plane_elems) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_pairs) return;
    const float2 v = reinterpret_cast<const float2*>(x)[i];
    uint32_t p0, p1, p2;
    x3_split2(v.x, v.y, p0, p1, p2);
    reinterpret_cast<uint32_t*>(planes)[i] = p0;
    reinterpret_cast<uint32_t*>(planes + plane_elems)[i] = p1;
    reinterpret_cast<uint32_t*>(planes + 2 * plane_elems)[i] = p2;
}

// the same from the TRANSPOSE: w [K, N] float (a weight stored input-major, as torch.matmul(x, w) wants it) -> planes [3][N][K];
// lanes run along n (coalesced reads of a row of w), each writes the pair (k, k + 1) of its plane rows
__global__ void k_split3_t(const float* __restrict__ w, unsigned short* __restrict__ planes, int K, int N) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)(K / 2) * N) return;
    const int n = (int)(i % N), kp = (int)(i / N);
    uint32_t p0, p1, p2;
    x3_split2(w[(size_t)(2 * kp) * N + n], w[(size_t)(2 * kp + 1) * N + n], p0, p1, p2);
    const size_t o = ((size_t)n * K + 2 * kp) / 2, plane_words = (size_t)N * K / 2;
    reinterpret_cast<uint32_t*>(planes)[o] = p0;
    reinterpret_cast<uint32_t*>(planes)[plane_words + o] = p1;
    reinterpret_cast<uint32_t*>(planes)[2 * plane_words + o] = p2;
}

// one float -> its three pieces (outer_step.h: the optimiser's kernel writes the planes of the weights it has just updated)
__device__ __forceinline__ void split_one(float x, unsigned short& q0, unsigned short& q1, unsigned short& q2) {
    uint32_t p0, p1, p2;
    x3_split2(x, 0.f, p0, p1, p2);
    q0 = (unsigned short)(p0 & 0xffffu); q1 = (unsigned short)(p1 & 0xffffu); q2 = (unsigned short)(p2 & 0xffffu);
}

struct Dense3Args {
    const float* A; int lda;                    // [M, K] activations, row stride lda (multiple of 4, 16-byte aligned rows)
    const unsigned short* Bp; size_t b_plane;   // pre-split weights: planes [3][N][K] (k_split3), plane stride in elements
    const float* bias;                          // [N] or null
    float* C; int ldc;                          // [M, N]
    int M, N, K;                                // K a multiple of 32; M, N arbitrary
};

extern __shared__ __attribute__((aligned(16))) unsigned short d3_lds[];

__global__ __launch_bounds__(D3_NT) void k_dense3(Dense3Args a) {
    constexpr int MI = 4, MJ = 2;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 2, wc = wv & 3;            // 2 x 4 waves of 64 x 32
    const int fi = lane & 15, fk = lane >> 4;
    const int tiles_n = (a.N + D3_TN - 1) / D3_TN;
    const int m0 = (blockIdx.x / tiles_n) * D3_TM, n0 = (blockIdx.x % tiles_n) * D3_TN;
    unsigned short* const As = d3_lds;                       // [2][3][PLANE]
    unsigned short* const Bs = d3_lds + 2 * 3 * D3_PLANE;    // [2][3][PLANE]

    // staging maps: row / column r = tid / 4 of the tile, k run 8 (tid % 4) .. + 7
    const int sr = tid >> 2, sk = (tid & 3) * 8;
    const bool a_ok = m0 + sr < a.M, b_ok = n0 + sr < a.N;
    const float* ap = a.A + (size_t)(a_ok ? m0 + sr : 0) * a.lda + sk;
    const unsigned short* bp = a.Bp + (size_t)(b_ok ? n0 + sr : 0) * a.K + sk;
    const int sdst = sr * X3_RS + sk;

    float4 ra0, ra1; uint4 rb0, rb1, rb2;
    auto fetch = [&](int k0) __attribute__((always_inline)) {
        ra0 = *reinterpret_cast<const float4*>(ap + k0); ra1 = *reinterpret_cast<const float4*>(ap + k0 + 4);
        rb0 = *reinterpret_cast<const uint4*>(bp + k0); rb1 = *reinterpret_cast<const uint4*>(bp + a.b_plane + k0);
        rb2 = *reinterpret_cast<const uint4*>(bp + 2 * a.b_plane + k0);
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
        // (rows / columns beyond M / N read row 0 instead - valid memory - and are staged like the others: a row of A only reaches its
        // own row of C, a column of B only its own column, and the epilogue stores neither)
        uint32_t p0[4], p1[4], p2[4];
        x3_split2(ra0.x, ra0.y, p0[0], p1[0], p2[0]); x3_split2(ra0.z, ra0.w, p0[1], p1[1], p2[1]);
        x3_split2(ra1.x, ra1.y, p0[2], p1[2], p2[2]); x3_split2(ra1.z, ra1.w, p0[3], p1[3], p2[3]);
        unsigned short* da = As + buf * 3 * D3_PLANE + sdst;
        *reinterpret_cast<uint4*>(da) = make_uint4(p0[0], p0[1], p0[2], p0[3]);
        *reinterpret_cast<uint4*>(da + D3_PLANE) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
        *reinterpret_cast<uint4*>(da + 2 * D3_PLANE) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
        unsigned short* db = Bs + buf * 3 * D3_PLANE + sdst;
        *reinterpret_cast<uint4*>(db) = rb0;
        *reinterpret_cast<uint4*>(db + D3_PLANE) = rb1;
        *reinterpret_cast<uint4*>(db + 2 * D3_PLANE) = rb2;
    };

    f32x4 acc[MI][MJ], small[MI][MJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) { acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; small[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    auto multiply = [&](int buf) __attribute__((always_inline)) {
        const unsigned short* Ab = As + buf * 3 * D3_PLANE + (wr * 64 + fi) * X3_RS + 8 * fk;
        const unsigned short* Bb = Bs + buf * 3 * D3_PLANE + (wc * 32 + fi) * X3_RS + 8 * fk;
        bf16x8 b0[MJ], b1[MJ], b2[MJ], ax[MI];
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            b0[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 16 * X3_RS);
            b1[j] = *reinterpret_cast<const bf16x8*>(Bb + D3_PLANE + j * 16 * X3_RS);
            b2[j] = *reinterpret_cast<const bf16x8*>(Bb + 2 * D3_PLANE + j * 16 * X3_RS);
        }
#if D3_EAGER_A
        bf16x8 a0[MI], a1[MI], a2[MI];
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            a0[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 16 * X3_RS);
            a1[i] = *reinterpret_cast<const bf16x8*>(Ab + D3_PLANE + i * 16 * X3_RS);
            a2[i] = *reinterpret_cast<const bf16x8*>(Ab + 2 * D3_PLANE + i * 16 * X3_RS);
        }
#define ADKF_D3_TERM(dst_, aq_, bq_)                                                                      \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < MJ; ++j)       \
        dst_[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(aq_[i], bq_[j], dst_[i][j], 0, 0, 0);
        ADKF_D3_TERM(small, a2, b0) ADKF_D3_TERM(small, a1, b1) ADKF_D3_TERM(small, a0, b2)
        ADKF_D3_TERM(small, a1, b0) ADKF_D3_TERM(small, a0, b1) ADKF_D3_TERM(acc, a0, b0)
        (void)ax;
#else
#define ADKF_D3_LOADA(q_) _Pragma("unroll") for (int i = 0; i < MI; ++i) ax[i] = *reinterpret_cast<const bf16x8*>(Ab + (q_) * D3_PLANE + i * 16 * X3_RS);
#define ADKF_D3_TERM(dst_, bq_)                                                                           \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < MJ; ++j)       \
        dst_[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[i], bq_[j], dst_[i][j], 0, 0, 0);
        ADKF_D3_LOADA(2) ADKF_D3_TERM(small, b0)                               // x2 y0
        ADKF_D3_LOADA(1) ADKF_D3_TERM(small, b1) ADKF_D3_TERM(small, b0)       // x1 y1, x1 y0
        ADKF_D3_LOADA(0) ADKF_D3_TERM(small, b2) ADKF_D3_TERM(small, b1) ADKF_D3_TERM(acc, b0)   // x0 y2, x0 y1, x0 y0
#undef ADKF_D3_LOADA
#endif
#undef ADKF_D3_TERM
#undef ADKF_D3_LOADA
    };

    const int nc = a.K / GK;
    fetch(0);
    stage(0);
    if (nc > 1) fetch(GK);
    __syncthreads();
    const bool stage_first = D3_PINGPONG ? wv < 4 : true;   // the two waves of a SIMD (wv, wv + 4) take the halves of an iteration in opposite order
    for (int c = 0; c < nc; ++c) {
        const int cur = c & 1;
        if (stage_first) {
            if (c + 1 < nc) { stage(cur ^ 1); if (c + 2 < nc && !(D3_ABLATE & 1)) fetch((c + 2) * GK); }
            if (!(D3_ABLATE & 4)) multiply(cur);
        } else {
            if (!(D3_ABLATE & 4)) multiply(cur);
            if (c + 1 < nc) { stage(cur ^ 1); if (c + 2 < nc && !(D3_ABLATE & 1)) fetch((c + 2) * GK); }
        }
        __syncthreads();
    }

    // epilogue: C/D map col = lane & 15, row = 4 (lane >> 4) + reg
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            const int gi0 = m0 + wr * 64 + i * 16 + fk * 4, gj = n0 + wc * 32 + j * 16 + fi;
            if (gj >= a.N) continue;
            if ((D3_ABLATE & 2) && acc[i][j][0] != 123.456f) continue;
            const float bv = a.bias ? a.bias[gj] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (gi0 + r < a.M) a.C[(size_t)(gi0 + r) * a.ldc + gj] = (acc[i][j][r] + small[i][j][r]) + bv;
        }
}

// ---- k_dense3_tn: the weight gradient dW[N, K] = G[M, N]^T X[M, K] (contraction over the ROWS of both operands) ----------------------
// Same tile, waves, product scheme and pipeline as k_dense3.  Both operands are activations (split on the fly) and both are strided
// along the contraction, so each lane takes a COLUMN (of G for the A image, of X for the B image) and eight consecutive rows of it per
// chunk: eight scalar loads, each a coalesced 256-byte row segment across the wave, and the lane holds exactly the k run its LDS row
// wants (one 16-byte store per plane; gemm_x3.h's column staging).  The contraction is long (all rows) and the output small, so it is
// cut into `splits` row ranges (blockIdx.y) whose partial products go to part[split][N][K]; k_dense3_reduce adds them in a fixed order
// (no atomics: bit-reproducible).  Rows beyond M are read clamped and contribute zeros.
struct Dense3TnArgs {
    const float* G; int ldg;      // [M, N]
    const float* X; int ldx;      // [M, K]
    float* part;                  // [splits, N, K]
    int M, N, K, rows_per_split;  // rows_per_split a multiple of 32
};

__global__ __launch_bounds__(D3_NT) void k_dense3_tn(Dense3TnArgs a) {
    constexpr int MI = 4, MJ = 2;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 2, wc = wv & 3;
    const int fi = lane & 15, fk = lane >> 4;
    const int tiles_k = (a.K + D3_TN - 1) / D3_TN;
    const int n0 = (blockIdx.x / tiles_k) * D3_TM, k0 = (blockIdx.x % tiles_k) * D3_TN;   // output tile: rows n0.. of dW, columns k0..
    const int r_begin = blockIdx.y * a.rows_per_split, r_end = min(a.M, r_begin + a.rows_per_split);
    unsigned short* const As = d3_lds;
    unsigned short* const Bs = d3_lds + 2 * 3 * D3_PLANE;

    const int sc = tid & 127, srun = (tid >> 7) * 8;     // this lane's column of the tile and its run of eight rows inside a chunk
    const float* gp = a.G + (n0 + sc < a.N ? n0 + sc : 0);
    const float* xp = a.X + (k0 + sc < a.K ? k0 + sc : 0);
    const int sdst = sc * X3_RS + srun;

    float rg[8], rx[8];
    auto fetch = [&](int row0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = row0 + srun + q;
            const int rc = r < r_end ? r : r_end - 1;
            const float g = gp[(size_t)rc * a.ldg], x = xp[(size_t)rc * a.ldx];
            rg[q] = r < r_end ? g : 0.f; rx[q] = r < r_end ? x : 0.f;
        }
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
        uint32_t p0[4], p1[4], p2[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) x3_split2(rg[2 * h], rg[2 * h + 1], p0[h], p1[h], p2[h]);
        unsigned short* da = As + buf * 3 * D3_PLANE + sdst;
        *reinterpret_cast<uint4*>(da) = make_uint4(p0[0], p0[1], p0[2], p0[3]);
        *reinterpret_cast<uint4*>(da + D3_PLANE) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
        *reinterpret_cast<uint4*>(da + 2 * D3_PLANE) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
#pragma unroll
        for (int h = 0; h < 4; ++h) x3_split2(rx[2 * h], rx[2 * h + 1], p0[h], p1[h], p2[h]);
        unsigned short* db = Bs + buf * 3 * D3_PLANE + sdst;
        *reinterpret_cast<uint4*>(db) = make_uint4(p0[0], p0[1], p0[2], p0[3]);
        *reinterpret_cast<uint4*>(db + D3_PLANE) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
        *reinterpret_cast<uint4*>(db + 2 * D3_PLANE) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
    };

    f32x4 acc[MI][MJ], small[MI][MJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) { acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; small[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    auto multiply = [&](int buf) __attribute__((always_inline)) {
        const unsigned short* Ab = As + buf * 3 * D3_PLANE + (wr * 64 + fi) * X3_RS + 8 * fk;
        const unsigned short* Bb = Bs + buf * 3 * D3_PLANE + (wc * 32 + fi) * X3_RS + 8 * fk;
        bf16x8 b0[MJ], b1[MJ], b2[MJ], ax[MI];
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            b0[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 16 * X3_RS);
            b1[j] = *reinterpret_cast<const bf16x8*>(Bb + D3_PLANE + j * 16 * X3_RS);
            b2[j] = *reinterpret_cast<const bf16x8*>(Bb + 2 * D3_PLANE + j * 16 * X3_RS);
        }
#define ADKF_D3_LOADA(q_) _Pragma("unroll") for (int i = 0; i < MI; ++i) ax[i] = *reinterpret_cast<const bf16x8*>(Ab + (q_) * D3_PLANE + i * 16 * X3_RS);
#define ADKF_D3_TERM(dst_, bq_)                                                                           \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < MJ; ++j)       \
        dst_[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[i], bq_[j], dst_[i][j], 0, 0, 0);
        ADKF_D3_LOADA(2) ADKF_D3_TERM(small, b0)
        ADKF_D3_LOADA(1) ADKF_D3_TERM(small, b1) ADKF_D3_TERM(small, b0)
        ADKF_D3_LOADA(0) ADKF_D3_TERM(small, b2) ADKF_D3_TERM(small, b1) ADKF_D3_TERM(acc, b0)
#undef ADKF_D3_TERM
#undef ADKF_D3_LOADA
    };

    const int nc = r_end > r_begin ? (r_end - r_begin + GK - 1) / GK : 0;
    if (nc > 0) {
        fetch(r_begin);
        stage(0);
        if (nc > 1) fetch(r_begin + GK);
        __syncthreads();
        const bool stage_first = wv < 4;
        for (int c = 0; c < nc; ++c) {
            const int cur = c & 1;
            if (stage_first) {
                if (c + 1 < nc) { stage(cur ^ 1); if (c + 2 < nc) fetch(r_begin + (c + 2) * GK); }
                multiply(cur);
            } else {
                multiply(cur);
                if (c + 1 < nc) { stage(cur ^ 1); if (c + 2 < nc) fetch(r_begin + (c + 2) * GK); }
            }
            __syncthreads();
        }
    }
    float* out = a.part + (size_t)blockIdx.y * a.N * a.K;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            const int gi0 = n0 + wr * 64 + i * 16 + fk * 4, gj = k0 + wc * 32 + j * 16 + fi;
            if (gj >= a.K) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (gi0 + r < a.N) out[(size_t)(gi0 + r) * a.K + gj] = acc[i][j][r] + small[i][j][r];
        }
}

// ---- the streaming forms (round 5, last session): short contractions and the weight gradient are bound by LOAD LATENCY, not by traffic ---
// At 65 536 x 256 x 256 (the C2 stand-in feature map) k_dense3 takes 74 us and k_dense3_tn 67.5: 1.8 TB/s of an 8 TB/s memory and a
// third of the matrix-pipe rate.  Both have ONE chunk of operand loads in flight per lane (16 - 32 KB per CU, 4 - 8 MB over the chip,
// where ~16 MB are needed to cover a 2 us trip to HBM at full rate): with eight chunks per tile, or every chunk's operands coming from
// HBM, each chunk pays most of a memory round trip (8 x 2 us + epilogue = the 18.5 us a workgroup takes).
//
#ifndef D3TN_ORDER
#define D3TN_ORDER 0
#endif
#ifndef D3SK_ORDER
#define D3SK_ORDER 2
#endif
#ifndef D3S_ABLATE   // diagnostics (tools/x3_stream_bench.hip): 1 no operand loads behind the prologue, 2 no stores of the result, 4 no MFMAs, 8 no LDS stores of the staging
#define D3S_ABLATE 0
#endif
// d3_multiply: the product of one staged chunk (both kernels below; the same instruction order as k_dense3's).
// SWAP: the MFMA takes the B fragment as its first operand, i.e. computes the transposed 16 x 16 block - the same products summed in the same
// order, but a lane then holds FOUR CONSECUTIVE COLUMNS of one output row (row = lane & 15, columns 4 (lane >> 4) + reg): one 16-byte store
// instead of four 4-byte ones.
template <bool SWAP = false>
__device__ __forceinline__ void d3_multiply(const unsigned short* Ab, const unsigned short* Bb, f32x4 (&acc)[4][2], f32x4 (&small)[4][2]) {
    constexpr int MI = 4, MJ = 2;
    bf16x8 b0[MJ], b1[MJ], b2[MJ], ax[MI];
#pragma unroll
    for (int j = 0; j < MJ; ++j) {
        b0[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 16 * X3_RS);
        b1[j] = *reinterpret_cast<const bf16x8*>(Bb + D3_PLANE + j * 16 * X3_RS);
        b2[j] = *reinterpret_cast<const bf16x8*>(Bb + 2 * D3_PLANE + j * 16 * X3_RS);
    }
#define ADKF_D3_LOADA(q_) _Pragma("unroll") for (int i = 0; i < MI; ++i) ax[i] = *reinterpret_cast<const bf16x8*>(Ab + (q_) * D3_PLANE + i * 16 * X3_RS);
#define ADKF_D3_TERM(dst_, bq_)                                                                           \
    _Pragma("unroll") for (int i = 0; i < MI; ++i) _Pragma("unroll") for (int j = 0; j < MJ; ++j)       \
        dst_[i][j] = SWAP ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(bq_[j], ax[i], dst_[i][j], 0, 0, 0)  \
                          : __builtin_amdgcn_mfma_f32_16x16x32_bf16(ax[i], bq_[j], dst_[i][j], 0, 0, 0);
    if (D3S_ABLATE & 4) return;
    ADKF_D3_LOADA(2) ADKF_D3_TERM(small, b0)
    ADKF_D3_LOADA(1) ADKF_D3_TERM(small, b1) ADKF_D3_TERM(small, b0)
    ADKF_D3_LOADA(0) ADKF_D3_TERM(small, b2) ADKF_D3_TERM(small, b1) ADKF_D3_TERM(acc, b0)
#undef ADKF_D3_TERM
#undef ADKF_D3_LOADA
}

// k_dense3_sk<NC>: C = A B^T (+ bias) for a SHORT contraction K = 32 NC (NC <= 8) and many rows.  Persistent: workgroup b walks the row
// tiles b, b + grid, ...; a row tile's WHOLE K extent of A sits in registers (NC x 8 floats per lane), is staged chunk by chunk for
// each column tile in turn, and - during the pass over the last column tile - every register set is refilled with the NEXT row tile's
// chunk as soon as it has been staged for the last time: A is read from HBM once, NC chunks (a whole tile pass) ahead of its use, and
// the chunk pipeline (double-buffered LDS, one barrier per chunk, ping-pong waves: k_dense3's) runs through tile boundaries without
// draining.  B (pre-split planes, small, L2-resident) keeps its one-chunk-ahead fetch.  The result of a tile goes out between two
// chunks of the pipeline (registers only).  Same products in the same order as k_dense3: bit-identical results.
template <int NC>
__global__ __launch_bounds__(D3_NT) void k_dense3_sk(Dense3Args a) {
    static_assert(NC >= 2 && NC <= 8, "K = 32 NC, 64 .. 256");
    constexpr int MI = 4, MJ = 2;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 2, wc = wv & 3;
    const int fi = lane & 15, fk = lane >> 4;
    const int tiles_m = (a.M + D3_TM - 1) / D3_TM, tiles_n = (a.N + D3_TN - 1) / D3_TN;
    unsigned short* const As = d3_lds;
    unsigned short* const Bs = d3_lds + 2 * 3 * D3_PLANE;
    const int sr = tid >> 2, sk = (tid & 3) * 8, sdst = sr * X3_RS + sk;
    const int frag_a = (wr * 64 + fi) * X3_RS + 8 * fk, frag_b = (wc * 32 + fi) * X3_RS + 8 * fk;

    int mt = blockIdx.x, nt = 0;
    if (mt >= tiles_m) return;
    float4 ra[NC][2]; uint4 rb0, rb1, rb2;
    // (rows / columns beyond M / N read row 0 instead - valid memory - as in k_dense3: they only reach outputs that are not stored)
    auto a_row = [&](int mt_) { const int r = mt_ * D3_TM + sr; return a.A + (size_t)(r < a.M ? r : 0) * a.lda + sk; };
    auto b_row = [&](int nt_) { const int r = nt_ * D3_TN + sr; return a.Bp + (size_t)(r < a.N ? r : 0) * a.K + sk; };
    auto fetch_a = [&](const float* ap, int c) __attribute__((always_inline)) {
        ra[c][0] = *reinterpret_cast<const float4*>(ap + c * GK); ra[c][1] = *reinterpret_cast<const float4*>(ap + c * GK + 4);
    };
    auto fetch_b = [&](const unsigned short* bp, int c) __attribute__((always_inline)) {
        rb0 = *reinterpret_cast<const uint4*>(bp + c * GK); rb1 = *reinterpret_cast<const uint4*>(bp + a.b_plane + c * GK);
        rb2 = *reinterpret_cast<const uint4*>(bp + 2 * a.b_plane + c * GK);
    };
    auto stage = [&](int buf, int c) __attribute__((always_inline)) {
        if (D3S_ABLATE & 8) return;
        uint32_t p0[4], p1[4], p2[4];
        x3_split2(ra[c][0].x, ra[c][0].y, p0[0], p1[0], p2[0]); x3_split2(ra[c][0].z, ra[c][0].w, p0[1], p1[1], p2[1]);
        x3_split2(ra[c][1].x, ra[c][1].y, p0[2], p1[2], p2[2]); x3_split2(ra[c][1].z, ra[c][1].w, p0[3], p1[3], p2[3]);
        unsigned short* da = As + buf * 3 * D3_PLANE + sdst;
        *reinterpret_cast<uint4*>(da) = make_uint4(p0[0], p0[1], p0[2], p0[3]);
        *reinterpret_cast<uint4*>(da + D3_PLANE) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
        *reinterpret_cast<uint4*>(da + 2 * D3_PLANE) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
        unsigned short* db = Bs + buf * 3 * D3_PLANE + sdst;
        *reinterpret_cast<uint4*>(db) = rb0;
        *reinterpret_cast<uint4*>(db + D3_PLANE) = rb1;
        *reinterpret_cast<uint4*>(db + 2 * D3_PLANE) = rb2;
    };

    const unsigned short* bp = b_row(0);
    {
        const float* ap = a_row(mt);
#pragma unroll
        for (int c = 0; c < NC; ++c) fetch_a(ap, c);
    }
    fetch_b(bp, 0);
    stage(0, 0);
    fetch_b(bp, 1);
    __syncthreads();
    // (measured at 65 536 x 256 x 256, tools/x3_stream_bench.hip: k_dense3's ping-pong order 61.7 us, everybody staging first 59.6, everybody
    // multiplying first 58.0 - with the operand loads long in flight there is no load latency left for a partner wave to cover)
    const bool stage_first = D3SK_ORDER == 0 ? wv < 4 : D3SK_ORDER == 1;   // 0: ping-pong, 1: everybody stages first, 2: everybody multiplies first
    const bool c_vec = !(a.ldc & 3) && !(reinterpret_cast<uintptr_t>(a.C) & 15);   // 16-byte stores of the result
    int p = 0;
    for (;;) {   // one output tile (mt, nt) per trip
        const bool last_n = nt + 1 == tiles_n;
        const int nt2 = last_n ? 0 : nt + 1, mt2 = last_n ? mt + (int)gridDim.x : mt;
        const bool have2 = mt2 < tiles_m;            // there is a tile behind this one
        const bool refill = last_n && have2;         // this pass is the last use of the row tile's registers
        const unsigned short* bp2 = b_row(nt2);
        const float* ap2 = a_row(have2 ? mt2 : mt);
        f32x4 acc[MI][MJ], small[MI][MJ];
#pragma unroll
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < MJ; ++j) { acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; small[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        if (refill && !(D3S_ABLATE & 1)) fetch_a(ap2, 0);   // (chunk 0 was staged for this tile in the previous trip)
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int c1 = (c + 1) % NC, c2 = (c + 2) % NC;
            const bool have_next = c + 1 < NC || have2;
            auto advance = [&]() __attribute__((always_inline)) {
                if (have_next) {
                    stage(p ^ 1, c1);
                    if (!(D3S_ABLATE & 1)) {
                        if (refill && c1 != 0) fetch_a(ap2, c1);
                        if (c + 2 < NC) fetch_b(bp, c2);
                        else if (have2) fetch_b(bp2, c2);
                    }
                }
            };
            if (stage_first) { advance(); d3_multiply<true>(As + p * 3 * D3_PLANE + frag_a, Bs + p * 3 * D3_PLANE + frag_b, acc, small); }
            else { d3_multiply<true>(As + p * 3 * D3_PLANE + frag_a, Bs + p * 3 * D3_PLANE + frag_b, acc, small); advance(); }
            __syncthreads();
            p ^= 1;
        }
        // the tile's result (transposed blocks: row = lane & 15, columns 4 (lane >> 4) + reg)
        const int m0 = mt * D3_TM, n0 = nt * D3_TN;
#pragma unroll
        for (int i = 0; i < MI; ++i) {
            const int gi = m0 + wr * 64 + i * 16 + fi;
            if (gi >= a.M) continue;
            float* crow = a.C + (size_t)gi * a.ldc;
#pragma unroll
            for (int j = 0; j < MJ; ++j) {
                const int gj0 = n0 + wc * 32 + j * 16 + fk * 4;
                if (gj0 >= a.N) continue;
                if ((D3S_ABLATE & 2) && acc[i][j][0] != 123.456f) continue;
                float o[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) o[r] = (acc[i][j][r] + small[i][j][r]) + ((a.bias && gj0 + r < a.N) ? a.bias[gj0 + r] : 0.f);
                if (c_vec && gj0 + 3 < a.N) *reinterpret_cast<float4*>(crow + gj0) = make_float4(o[0], o[1], o[2], o[3]);
                else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) if (gj0 + r < a.N) crow[gj0 + r] = o[r];
                }
            }
        }
        if (!have2) break;
        mt = mt2; nt = nt2; bp = bp2;
    }
}

// k_dense3_tnd<DEPTH>: k_dense3_tn with DEPTH chunks of operand loads in flight per lane (register sets used round robin; the chunk loop
// is unrolled DEPTH times so that the sets are static), and an XCD-aware order of the workgroups: the output tiles of one row range
// land on ONE XCD next to each other in time, so the second reader of an operand half finds it in that XCD's L2.  Products, chunk
// order and row ranges are k_dense3_tn's: bit-identical partial sums.
template <int DEPTH>
__global__ __launch_bounds__(D3_NT) void k_dense3_tnd(Dense3TnArgs a, int tiles, int splits) {
    constexpr int MI = 4, MJ = 2;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int wr = wv >> 2, wc = wv & 3;
    const int fi = lane & 15, fk = lane >> 4;
    const int tiles_k = (a.K + D3_TN - 1) / D3_TN;
    // workgroup id -> (tile, row range): ids b, b + 8, b + 16, ... share an XCD (round-robin dispatch); a multiple-of-8 prefix of the
    // grid is renumbered XCD-major so that consecutive logical indices - the tiles of one row range - share an XCD
    const bool remap = splits > 0;   // (splits < 0: dispatch order, for A/B runs)
    if (splits < 0) splits = -splits;
    const int total = tiles * splits, per = total >> 3, b = blockIdx.x;
    const int logical = (remap && b < 8 * per) ? (b & 7) * per + (b >> 3) : b;
    const int tile = logical % tiles, split = logical / tiles;
    const int n0 = (tile / tiles_k) * D3_TM, k0 = (tile % tiles_k) * D3_TN;
    const int r_begin = split * a.rows_per_split, r_end = min(a.M, r_begin + a.rows_per_split);
    unsigned short* const As = d3_lds;
    unsigned short* const Bs = d3_lds + 2 * 3 * D3_PLANE;
    const int sc = tid & 127, srun = (tid >> 7) * 8;
    const float* gp = a.G + (n0 + sc < a.N ? n0 + sc : 0);
    const float* xp = a.X + (k0 + sc < a.K ? k0 + sc : 0);
    const int sdst = sc * X3_RS + srun;
    const int frag_a = (wr * 64 + fi) * X3_RS + 8 * fk, frag_b = (wc * 32 + fi) * X3_RS + 8 * fk;

    float rg[DEPTH][8], rx[DEPTH][8];
    auto fetch = [&](int set, int row0) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int r = row0 + srun + q;
            const int rc = r < r_end ? r : r_end - 1;
            const float g = gp[(size_t)rc * a.ldg], x = xp[(size_t)rc * a.ldx];
            rg[set][q] = r < r_end ? g : 0.f; rx[set][q] = r < r_end ? x : 0.f;
        }
    };
    auto stage = [&](int buf, int set) __attribute__((always_inline)) {
        if (D3S_ABLATE & 8) return;
        uint32_t p0[4], p1[4], p2[4];
#pragma unroll
        for (int h = 0; h < 4; ++h) x3_split2(rg[set][2 * h], rg[set][2 * h + 1], p0[h], p1[h], p2[h]);
        unsigned short* da = As + buf * 3 * D3_PLANE + sdst;
        *reinterpret_cast<uint4*>(da) = make_uint4(p0[0], p0[1], p0[2], p0[3]);
        *reinterpret_cast<uint4*>(da + D3_PLANE) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
        *reinterpret_cast<uint4*>(da + 2 * D3_PLANE) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
#pragma unroll
        for (int h = 0; h < 4; ++h) x3_split2(rx[set][2 * h], rx[set][2 * h + 1], p0[h], p1[h], p2[h]);
        unsigned short* db = Bs + buf * 3 * D3_PLANE + sdst;
        *reinterpret_cast<uint4*>(db) = make_uint4(p0[0], p0[1], p0[2], p0[3]);
        *reinterpret_cast<uint4*>(db + D3_PLANE) = make_uint4(p1[0], p1[1], p1[2], p1[3]);
        *reinterpret_cast<uint4*>(db + 2 * D3_PLANE) = make_uint4(p2[0], p2[1], p2[2], p2[3]);
    };

    f32x4 acc[MI][MJ], small[MI][MJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) { acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; small[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    const int nc = r_end > r_begin ? (r_end - r_begin + GK - 1) / GK : 0;
    if (nc > 0) {
        // chunk q travels in set q % DEPTH; chunks 1 .. DEPTH are in flight when the loop starts
#pragma unroll
        for (int q = 0; q < DEPTH; ++q) if (q < nc) fetch(q, r_begin + q * GK);
        stage(0, 0);
        if (DEPTH < nc) fetch(0, r_begin + DEPTH * GK);
        __syncthreads();
        const bool stage_first = D3TN_ORDER == 0 ? wv < 4 : D3TN_ORDER == 1;   // as D3SK_ORDER
        for (int c0 = 0; c0 < nc; c0 += DEPTH) {
#pragma unroll
            for (int u = 0; u < DEPTH; ++u) {
                const int c = c0 + u;
                if (c < nc) {   // (uniform over the workgroup: the barrier below is reached by everybody or nobody)
                    const int cur = (DEPTH & 1) ? (c & 1) : (u & 1), set1 = (u + 1) % DEPTH;
                    auto advance = [&]() __attribute__((always_inline)) {
                        if (c + 1 < nc) { stage(cur ^ 1, set1); if (c + 1 + DEPTH < nc && !(D3S_ABLATE & 1)) fetch(set1, r_begin + (c + 1 + DEPTH) * GK); }
                    };
                    if (stage_first) { advance(); d3_multiply(As + cur * 3 * D3_PLANE + frag_a, Bs + cur * 3 * D3_PLANE + frag_b, acc, small); }
                    else { d3_multiply(As + cur * 3 * D3_PLANE + frag_a, Bs + cur * 3 * D3_PLANE + frag_b, acc, small); advance(); }
                    __syncthreads();
                }
            }
        }
    }
    float* out = a.part + (size_t)split * a.N * a.K;
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < MJ; ++j) {
            const int gi0 = n0 + wr * 64 + i * 16 + fk * 4, gj = k0 + wc * 32 + j * 16 + fi;
            if (gj >= a.K) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (gi0 + r < a.N) out[(size_t)(gi0 + r) * a.K + gj] = acc[i][j][r] + small[i][j][r];
        }
}

// dW = part[0] + part[1] + ... in that order
__global__ void k_dense3_reduce(const float* __restrict__ part, float* __restrict__ dw, size_t n, int splits) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // (the ORDER of the additions is fixed; the loads are independent and issued eight at a time - one by one, each waiting for the
    // one before, the 64 ranges of the C2 feature map took 16.7 us)
    float s = part[i];
    int q = 1;
    for (; q + 8 <= splits; q += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(q + u) * n + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; q < splits; ++q) s += part[(size_t)q * n + i];
    dw[i] = s;
}

}  // namespace adkf
