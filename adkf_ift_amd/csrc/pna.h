// Fused aggregation of the relational multi-aggregation message passing (fs_mol/modules/gnn.py:197-265,
// RelationalMultiAggrMP._aggregate_messages): for every target node the SUM, MEAN, STD and MAX of its incoming
// messages in one kernel (and one for the backward), instead of three index_add, a gather, a scatter-amax and ten
// element-wise passes over the [E, towers, 3m] message tensor per layer.
//
// msgs   [E, H, 3m]  post-ReLU messages of all edge types: per tower (sum-part | mean/std-part | max-part)
// perm   [E]         message ids sorted by target node (stable), rowptr [V + 1] the segments
// agg    [V, H, 4m]  (sum | mean | std | max); empty segments give 0 (torch_scatter's convention)
// argmax [V, H, m]   message id of the maximum (-1: empty) - first maximum in segment order
//   std = sqrt(sum_e (relu(b_e^2 - mean^2) + 1e-7))   (gnn.py:213-216)
// One workgroup per target node, one thread per (tower, feature): rows are read as contiguous 4m-byte segments.
#pragma once
#include "device_utils.h"

namespace adkf {

constexpr float PNA_SMALL = 1e-7f;

struct PnaArgs {
    const float* msgs; const int64_t* perm; const int64_t* rowptr;
    float* agg; int32_t* argmax;
    const float* d_agg; float* d_msgs;   // backward only
    int V, H, m;
    int relu_mask;   // backward: the messages are ReLU outputs and d_msgs is to be the gradient IN FRONT of that ReLU (zero where msgs <= 0)
};

__global__ __launch_bounds__(256) void k_pna_fwd(PnaArgs a) {
    const int v = blockIdx.x, H = a.H, m = a.m;
    const int64_t p0 = a.rowptr[v], p1 = a.rowptr[v + 1];
    const float deg = (float)(p1 - p0);
    for (int idx = threadIdx.x; idx < H * m; idx += blockDim.x) {
        const int h = idx / m, f = idx - h * m;
        // mean and deviations in float64: the products b_e^2 and mean^2 of float32 numbers are exact there, so the difference
        // under the relu - which cancels to ~1e-7 b^2 for nearly equal messages, where the reference's std has slope
        // 1 / (2 sqrt(1e-7)) = 1581 - carries no rounding of this kernel's own (float32: eps b^2, i.e. as large as the 1e-7
        // floor itself); what is left is the float32 rounding of the messages that come in
        float s = 0.f, mx = -INFINITY;
        double b = 0.0;
        int am = -1;
        for (int64_t p = p0; p < p1; ++p) {
            const int64_t e = a.perm[p];
            const float* row = a.msgs + ((size_t)e * H + h) * 3 * m;
            s += row[f];
            b += (double)row[m + f];
            const float c = row[2 * m + f];
            if (c > mx) { mx = c; am = (int)e; }
        }
        const double mean = p1 > p0 ? b / (double)deg : 0.0;
        double dev = 0.0;
        for (int64_t p = p0; p < p1; ++p) {
            const double bb = (double)a.msgs[((size_t)a.perm[p] * H + h) * 3 * m + m + f];
            dev += fmax(bb * bb - mean * mean, 0.0) + (double)PNA_SMALL;
        }
        float* out = a.agg + ((size_t)v * H + h) * 4 * m;
        out[f] = s;
        out[m + f] = (float)mean;
        out[2 * m + f] = (float)sqrt(dev);
        out[3 * m + f] = p1 > p0 ? mx : 0.f;
        a.argmax[((size_t)v * H + h) * m + f] = am;
    }
}

// d msgs from d agg.  With dev_e = relu(b_e^2 - mean^2) + eps, std = sqrt(sum dev_e):
//   d b_e = [b_e^2 > mean^2] 2 b_e g  +  (d_mean - 2 mean g sum_e' [b_e'^2 > mean^2]) / deg,   g = d_std / (2 std)
//   d a_e = d_sum,   d c_e = d_max at the arg-max message, 0 elsewhere.
__global__ __launch_bounds__(256) void k_pna_bwd(PnaArgs a) {
    const int v = blockIdx.x, H = a.H, m = a.m;
    const int64_t p0 = a.rowptr[v], p1 = a.rowptr[v + 1];
    if (p1 <= p0) return;
    const float deg = (float)(p1 - p0);
    for (int idx = threadIdx.x; idx < H * m; idx += blockDim.x) {
        const int h = idx / m, f = idx - h * m;
        const float* ag = a.agg + ((size_t)v * H + h) * 4 * m;
        const float* dg = a.d_agg + ((size_t)v * H + h) * 4 * m;
        const float d_sum = dg[f], d_mean = dg[m + f], d_std = dg[2 * m + f], d_max = dg[3 * m + f];
        const int am = a.argmax[((size_t)v * H + h) * m + f];
        // mean, the indicators [b_e^2 > mean^2] and std again in float64, exactly as the forward formed them (the stored mean
        // and std are rounded to float32: an indicator evaluated with THAT mean can differ from the forward's)
        double bsum = 0.0;
        for (int64_t p = p0; p < p1; ++p) bsum += (double)a.msgs[((size_t)a.perm[p] * H + h) * 3 * m + m + f];
        const double mean = bsum / (double)deg;
        double dev = 0.0, cnt = 0.0;
        for (int64_t p = p0; p < p1; ++p) {
            const double bb = (double)a.msgs[((size_t)a.perm[p] * H + h) * 3 * m + m + f];
            const double x = bb * bb - mean * mean;
            dev += fmax(x, 0.0) + (double)PNA_SMALL;
            cnt += x > 0.0 ? 1.0 : 0.0;
        }
        const double g = (double)d_std / (2.0 * sqrt(dev));
        const double via_mean = ((double)d_mean - 2.0 * mean * g * cnt) / (double)deg;
        for (int64_t p = p0; p < p1; ++p) {
            const int64_t e = a.perm[p];
            const size_t o = ((size_t)e * H + h) * 3 * m;
            const double bb = (double)a.msgs[o + m + f];
            float da = d_sum, db = (float)(((bb * bb > mean * mean) ? 2.0 * bb * g : 0.0) + via_mean), dc = ((int)e == am) ? d_max : 0.f;
            if (a.relu_mask) {   // the consumers (ProbMsgBwdX / BwdW, k_msg_dbias) then read ONE tensor instead of gradient + mask
                da = a.msgs[o + f] > 0.f ? da : 0.f;
                db = bb > 0.0 ? db : 0.f;
                dc = a.msgs[o + 2 * m + f] > 0.f ? dc : 0.f;
            }
            a.d_msgs[o + f] = da;
            a.d_msgs[o + m + f] = db;
            a.d_msgs[o + 2 * m + f] = dc;
        }
    }
}

}  // namespace adkf

// ---------------------------------------------------------------------------------------------------------------------
// Message functions of one edge type for all towers (fs_mol/modules/gnn.py:95-148, depth-1 message MLP):
//     msgs[e, h, :] = relu( cat(x[src_e, h, :], x[tgt_e, h, :]) W[h] + b[h] )
// as ONE batched MFMA GEMM whose A operand is gathered on the fly (no [E, H, 2 in] concatenation, no separate bias /
// ReLU passes), and the two backward products with the ReLU mask fused into their operand loads.  The backward is
// REPRODUCIBLE to the bit (no floating-point atomics; round 4 - the reference's own scatter_add is not, which is why two
// runs of a float32 inner fit on "identical" features used to differ by 1e-4):
//     d cat[e, h, :] = (d msgs . [msgs > 0]) W[h]^T        written once per edge ([E_all, H, 2 in]); d x[v] is then GATHERED
//                                                          over v's outgoing (first half) and incoming (second half) edges in
//                                                          CSR order by k_msg_dx - every element of d x written exactly once
//     d W[h]  = sum over fixed chunks of edges of cat(x)^T (d msgs . [msgs > 0]): one partial per chunk, summed IN ORDER by
//     d b[h]    k_msg_reduce (the same for the bias partials of k_msg_dbias); the chunking depends on E alone
// "task" of k_bgemm = tower (forward, d cat) or (tower, edge chunk) (d W).  ALL edge types go through ONE launch of each kind
// (round 4: the flat tile / chunk index is mapped to its edge type by the functor; three launches per kind left the two rare bond
// types of a molecule - a few thousand edges - with a handful of workgroups each).
// ---------------------------------------------------------------------------------------------------------------------
#include "problems.h"
namespace adkf {

constexpr int MSG_MAX_ET = 4;

struct MsgEt {
    const int64_t *src, *tgt;  // [E]
    const float* W;            // [H, 2 in, out]
    const float* bias;         // [H, out]
    float *dW, *db;            // backward outputs of this edge type
    int E;
    int e_off;                 // first row of this edge type in msgs / d_msgs / dcat
    int tile0;                 // first flat tile of this edge type in the launch at hand (forward / d cat)
    int split0, chunk;         // first chunk of this edge type among all chunks; edges per chunk
};

struct MsgArgs {
    const float* x;            // [V, H, in]
    float* msgs;               // [E_all, H, out]; backward: null = d_msgs is already the gradient in front of the ReLU (PnaArgs::relu_mask)
    const float* d_msgs;       // backward
    float* dcat;               // [E_all, H, 2 in]
    float* part;               // [nsplit_all, H * 2 in * out + H * out]: per-chunk partials of d W | d b
    int H, in, out, n_et, nsplit_all;
    bool vec;
    MsgEt e0, e1, e2, e3;      // (four named rows, not an array: see msg_et)
};
inline MsgEt& msg_row(MsgArgs& m, int q) { return q == 0 ? m.e0 : q == 1 ? m.e1 : q == 2 ? m.e2 : m.e3; }   // host side

// Edges per split of the d W / d b products: a function of E ALONE (at most MSG_SPLITS partials, at least 512 edges each,
// a multiple of the GEMM's K chunk), so the summation order - hence the bits - does not depend on anything else.
constexpr int MSG_SPLITS = 64;
inline int msg_chunk(int E) {
    const int c = (E + MSG_SPLITS - 1) / MSG_SPLITS;
    const int r = ((c + 31) / 32) * 32;
    return r < 512 ? 512 : r;
}
inline int msg_nsplit(int E) { if (E <= 0) return 0; const int c = msg_chunk(E); return (E + c - 1) / c; }
__host__ __device__ inline size_t msg_part_stride(int H, int in, int out) { return (size_t)H * 2 * in * out + (size_t)H * out; }

// Edge type ei's row of the table.  The rows are four named members and the lookup a chain of selects: an array in the functor,
// indexed at run time or not, made hipcc keep the whole functor in private memory - every access in the operand loops then went
// through scratch and the merged forward launch ran 4 x slower than the three separate ones.
__device__ __forceinline__ MsgEt msg_et(const MsgArgs& m, int ei) {   // (whole-row copies in separate branches, like ProbDistMulti::select)
    MsgEt e;
    if (ei == 0) { e = m.e0; }
    else if (ei == 1) { e = m.e1; }
    else if (ei == 2) { e = m.e2; }
    else { e = m.e3; }
    return e;
}
__device__ __forceinline__ int msg_find_tile(const MsgArgs& m, int idx) {    // the edge type whose tile range holds idx
    int ei = 0;
    if (m.n_et > 1 && idx >= m.e1.tile0) ei = 1;
    if (m.n_et > 2 && idx >= m.e2.tile0) ei = 2;
    if (m.n_et > 3 && idx >= m.e3.tile0) ei = 3;
    return ei;
}
__device__ __forceinline__ int msg_find_split(const MsgArgs& m, int idx) {   // ... whose chunk range holds idx
    int ei = 0;
    if (m.n_et > 1 && idx >= m.e1.split0) ei = 1;
    if (m.n_et > 2 && idx >= m.e2.split0) ei = 2;
    if (m.n_et > 3 && idx >= m.e3.split0) ei = 3;
    return ei;
}

// d bias partial of one chunk: 64 columns x 4 row groups per workgroup (grid: ceil(H * out / 64) x nsplit_all), eight loads in
// flight per thread; the four row-group sums are combined in a fixed order
__global__ __launch_bounds__(256) void k_msg_dbias(MsgArgs m) {
    __shared__ float part[4][64];
    const int cl = threadIdx.x & 63, g = threadIdx.x >> 6, c = blockIdx.x * 64 + cl, width = m.H * m.out;
    const MsgEt et = msg_et(m, msg_find_split(m, (int)blockIdx.y));
    const int e0 = ((int)blockIdx.y - et.split0) * et.chunk, e1 = min(et.E, e0 + et.chunk);
    float s = 0.f;
    if (c < width) {
        int e = e0 + g;
        for (; e + 28 < e1; e += 32) {
            float dv[8], mv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const size_t o = (size_t)(et.e_off + e + 4 * u) * width + c;
                dv[u] = m.d_msgs[o]; mv[u] = m.msgs ? m.msgs[o] : 1.f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) s += mv[u] > 0.f ? dv[u] : 0.f;
        }
        for (; e < e1; e += 4) {
            const size_t o = (size_t)(et.e_off + e) * width + c;
            s += (!m.msgs || m.msgs[o] > 0.f) ? m.d_msgs[o] : 0.f;
        }
    }
    part[g][cl] = s;
    __syncthreads();
    if (g == 0 && c < width)
        m.part[(size_t)blockIdx.y * msg_part_stride(m.H, m.in, m.out) + (size_t)m.H * 2 * m.in * m.out + c] =
            (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]);
}

// out[i] = part[first][i] + part[first + 1][i] + ... (in this order) for the n_w elements of d W and the n_b of d b of every edge
// type (blockIdx.y); an edge type without edges gets exact zeros
__global__ __launch_bounds__(256) void k_msg_reduce(MsgArgs m) {
    const MsgEt et = msg_et(m, (int)blockIdx.y);
    const int n_w = m.H * 2 * m.in * m.out, n_b = m.H * m.out;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n_w + n_b) return;
    const size_t stride = msg_part_stride(m.H, m.in, m.out);
    const int nsplit = (et.E + et.chunk - 1) / et.chunk;
    const float* part = m.part + (size_t)et.split0 * stride;
    float s = 0.f;
    int p = 0;
    for (; p + 8 <= nsplit; p += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(p + u) * stride + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; p < nsplit; ++p) s += part[(size_t)p * stride + i];
    if (i < n_w) et.dW[i] = s; else et.db[i - n_w] = s;
}

// d x[v, h, :] = sum over v's outgoing edges of d cat[e, h, 0:in]  +  sum over its incoming edges of d cat[e, h, in:2 in],
// each list in CSR order (perm_s / rowptr_s by source, perm_t / rowptr_t by target, over the concatenated edge list of all
// edge types).  One thread per (node, column); a workgroup covers 256 / (H in) nodes.
struct MsgDxArgs {
    const float* dcat; const int64_t *perm_s, *rowptr_s, *perm_t, *rowptr_t; float* dx; int V, H, in;
};
__global__ __launch_bounds__(256) void k_msg_dx(MsgDxArgs a) {
    const int width = a.H * a.in;
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long v = gid / width;
    if (v >= a.V) return;
    const int c = (int)(gid - v * width), h = c / a.in, f = c - h * a.in;
    const size_t col = (size_t)h * 2 * a.in + f, ld = (size_t)a.H * 2 * a.in;
    float s = 0.f;
    for (int64_t p = a.rowptr_s[v], p1 = a.rowptr_s[v + 1]; p < p1; ++p) s += a.dcat[(size_t)a.perm_s[p] * ld + col];
    for (int64_t p = a.rowptr_t[v], p1 = a.rowptr_t[v + 1]; p < p1; ++p) s += a.dcat[(size_t)a.perm_t[p] * ld + col + a.in];
    a.dx[(size_t)v * width + c] = s;
}

// Inside the GEMM functors (kernel arguments that select() / setup() write to) a row is picked FIELD BY FIELD in explicit branches:
// a whole-row copy through msg_et() made hipcc keep the entire functor in private memory - every access of the operand loops went
// through scratch and the merged forward launch ran 4 x slower than three separate ones (tools/: bisected on the code object's
// private_segment_fixed_size).
#define MSG_ROW(ei_, stmt_) do { if ((ei_) == 0) { const MsgEt& r_ = m.e0; stmt_; } else if ((ei_) == 1) { const MsgEt& r_ = m.e1; stmt_; } \
                                 else if ((ei_) == 2) { const MsgEt& r_ = m.e2; stmt_; } else { const MsgEt& r_ = m.e3; stmt_; } } while (0)

// flat tile -> (edge type, tile inside it): the edge types' tiles are laid side by side (gemm.h: select())
__device__ __forceinline__ int msg_select(const MsgArgs& m, int& tile) {
    int ei = 0, t0 = 0;
    if (m.n_et > 1 && tile >= m.e1.tile0) { ei = 1; t0 = m.e1.tile0; }
    if (m.n_et > 2 && tile >= m.e2.tile0) { ei = 2; t0 = m.e2.tile0; }
    if (m.n_et > 3 && tile >= m.e3.tile0) { ei = 3; t0 = m.e3.tile0; }
    tile -= t0;
    return ei;
}

struct ProbMsgFwd {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = false;
    static constexpr int NRED = 0;
    MsgArgs m; int h, ei; bool vec;
    const int64_t *src, *tgt; const float *W, *bias; int E, e_off;
    __device__ void select(int& tile, int& tiles_n) { ei = msg_select(m, tile); tiles_n = (m.out + GT - 1) / GT; }
    __device__ bool setup(int t) {
        h = t; vec = m.vec;
        MSG_ROW(ei, (src = r_.src, tgt = r_.tgt, W = r_.W, bias = r_.bias, E = r_.E, e_off = r_.e_off));
        return E > 0;
    }
    __device__ int M() const { return E; } __device__ int N() const { return m.out; } __device__ int K() const { return 2 * m.in; }
    __device__ const float* arow(int i, int k) const {
        const int64_t node = k < m.in ? src[i] : tgt[i];
        return m.x + ((size_t)node * m.H + h) * m.in + (k < m.in ? k : k - m.in);
    }
    __device__ float a(int i, int k) const { return *arow(i, k); }
    __device__ float b(int k, int j) const { return W[((size_t)h * 2 * m.in + k) * m.out + j]; }
    __device__ void a4(int i, int k, float (&v)[4]) const { ld4(arow(i, k), v); }   // in % 4 == 0: a group never straddles the halves
    __device__ void b4(int k, int j, float (&v)[4]) const { ld4(W + ((size_t)h * 2 * m.in + k) * m.out + j, v); }
    __device__ void epi(int i, int j, float acc, float*) const {
        m.msgs[((size_t)(e_off + i) * m.H + h) * m.out + j] = fmaxf(acc + bias[h * m.out + j], 0.f);
    }
    __device__ void store_red(int, const float*) const {}
};

struct ProbMsgBwdX {
    static constexpr bool A_KCONTIG = true, B_KCONTIG = true;
    static constexpr int NRED = 0;
    MsgArgs m; int h, ei; bool vec;
    const float* W; int E, e_off;
    __device__ void select(int& tile, int& tiles_n) { ei = msg_select(m, tile); tiles_n = (2 * m.in + GT - 1) / GT; }
    __device__ bool setup(int t) {
        h = t; vec = m.vec;
        MSG_ROW(ei, (W = r_.W, E = r_.E, e_off = r_.e_off));
        return E > 0;
    }
    __device__ int M() const { return E; } __device__ int N() const { return 2 * m.in; } __device__ int K() const { return m.out; }
    __device__ float a(int i, int k) const {
        const size_t o = ((size_t)(e_off + i) * m.H + h) * m.out + k;
        return (!m.msgs || m.msgs[o] > 0.f) ? m.d_msgs[o] : 0.f;
    }
    __device__ float b(int k, int j) const { return W[((size_t)h * 2 * m.in + j) * m.out + k]; }
    __device__ void a4(int i, int k, float (&v)[4]) const {
        const size_t o = ((size_t)(e_off + i) * m.H + h) * m.out + k;
        ld4(m.d_msgs + o, v);
        if (m.msgs) {
            float ms[4];
            ld4(m.msgs + o, ms);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = ms[q] > 0.f ? v[q] : 0.f;
        }
    }
    __device__ void b4(int k, int j, float (&v)[4]) const { ld4(W + ((size_t)h * 2 * m.in + j) * m.out + k, v); }
    __device__ void epi(int i, int j, float acc, float*) const {
        m.dcat[((size_t)(e_off + i) * m.H + h) * 2 * m.in + j] = acc;
    }
    __device__ void store_red(int, const float*) const {}
};

struct ProbMsgBwdW {
    static constexpr bool A_KCONTIG = false, B_KCONTIG = false;
    static constexpr int NRED = 0;
    MsgArgs m; int h, sp, e0, len; bool vec;
    const int64_t *src, *tgt; int e_off;
    __device__ bool setup(int t) {
        h = t / m.nsplit_all; sp = t % m.nsplit_all; vec = m.vec;
        const int ei = msg_find_split(m, sp);
        int split0, chunk, E;
        MSG_ROW(ei, (src = r_.src, tgt = r_.tgt, e_off = r_.e_off, split0 = r_.split0, chunk = r_.chunk, E = r_.E));
        e0 = (sp - split0) * chunk; len = min(chunk, E - e0);
        return len > 0;
    }
    __device__ int M() const { return 2 * m.in; } __device__ int N() const { return m.out; } __device__ int K() const { return len; }
    __device__ const float* arow(int i, int k) const {
        const int64_t node = i < m.in ? src[e0 + k] : tgt[e0 + k];
        return m.x + ((size_t)node * m.H + h) * m.in + (i < m.in ? i : i - m.in);
    }
    __device__ float a(int i, int k) const { return *arow(i, k); }
    __device__ float b(int k, int j) const {
        const size_t o = ((size_t)(e_off + e0 + k) * m.H + h) * m.out + j;
        return (!m.msgs || m.msgs[o] > 0.f) ? m.d_msgs[o] : 0.f;
    }
    __device__ void a4(int i, int k, float (&v)[4]) const { ld4(arow(i, k), v); }
    __device__ void b4(int k, int j, float (&v)[4]) const {
        const size_t o = ((size_t)(e_off + e0 + k) * m.H + h) * m.out + j;
        ld4(m.d_msgs + o, v);
        if (m.msgs) {
            float ms[4];
            ld4(m.msgs + o, ms);
#pragma unroll
            for (int q = 0; q < 4; ++q) v[q] = ms[q] > 0.f ? v[q] : 0.f;
        }
    }
    __device__ void epi(int i, int j, float acc, float*) const {
        m.part[(size_t)sp * msg_part_stride(m.H, m.in, m.out) + ((size_t)h * 2 * m.in + i) * m.out + j] = acc;
    }
    __device__ void store_red(int, const float*) const {}
};

}  // namespace adkf
