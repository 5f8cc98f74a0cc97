// Per-graph pooling of the combined graph read-out (fs_mol/modules/graph_readout.py:119-177: weighted-mean head :180-259
// with scatter_softmax weights, weighted-sum head with sigmoid weights, max pooler :262-296), the part between the node-level
// MLPs and the per-graph combination layers:
//     w_mean[v, h] = softmax over the nodes of v's graph of s_mean[., h]          (torch_scatter.scatter_softmax)
//     g_mean[g, h, :] = sum_v w_mean[v, h] v_mean[v, h, :]                         (index_add_ in the reference)
//     w_sum[v, h]  = sigmoid(s_sum[v, h]);   g_sum[g, h, :] = sum_v w_sum[v, h] v_sum[v, h, :]
//     g_max[g, :]  = max_v emb[v, :]   (0 for a graph without nodes, torch_scatter's convention; argmax = first maximum)
// as ONE kernel forward and ONE backward, every sum taken over the graph's node list in a fixed order - the reference's (and
// PyTorch's) index_add_ / gather-backward are float atomics on a GPU, whose summation order changes from run to run (round 3's
// red test: the converged float32 inner fit turned that 1e-7 into 1.7e-4 on f_out).  The backward has no scatter at all: every
// output element belongs to one node and is written once.
//   perm [V]: node ids sorted by graph (stable), rowptr [G + 1] the segments; n2g [V] the graph of each node.
#pragma once
#include "device_utils.h"

namespace adkf {

struct ReadoutArgs {
    const float *s_mean, *v_mean, *s_sum, *v_sum, *emb;   // [V, nh], [V, nh hd], [V, nh], [V, nh hd], [V, D]
    const int64_t *perm, *rowptr, *n2g;
    float *w_mean, *w_sum;                                 // [V, nh] (forward: out, backward: in)
    float *g_mean, *g_sum, *g_max;                         // [G, nh hd], [G, nh hd], [G, D]
    int32_t* argmax;                                       // [G, D] node id of the maximum, -1: empty graph
    const float *dg_mean, *dg_sum, *dg_max;                // backward
    float *d_s_mean, *d_v_mean, *d_s_sum, *d_v_sum, *d_emb;
    int V, G, nh, hd, D;
};

constexpr int READOUT_MAX_HEADS = 64;

// one workgroup per graph
__global__ __launch_bounds__(256) void k_readout_fwd(ReadoutArgs a) {
    __shared__ float mx_s[READOUT_MAX_HEADS], den_s[READOUT_MAX_HEADS];
    const int g = blockIdx.x, tid = threadIdx.x, nh = a.nh, HD = a.nh * a.hd;
    const int64_t p0 = a.rowptr[g], p1 = a.rowptr[g + 1];
    const int n = (int)(p1 - p0);
    if (tid < nh) {
        float mx = -INFINITY;
        for (int64_t p = p0; p < p1; ++p) mx = fmaxf(mx, a.s_mean[(size_t)a.perm[p] * nh + tid]);
        float den = 0.f;
        for (int64_t p = p0; p < p1; ++p) den += expf(a.s_mean[(size_t)a.perm[p] * nh + tid] - mx);
        mx_s[tid] = mx; den_s[tid] = den;
    }
    __syncthreads();
    for (int idx = tid; idx < n * nh; idx += 256) {
        const int pl = idx / nh, h = idx - pl * nh;
        const size_t o = (size_t)a.perm[p0 + pl] * nh + h;
        a.w_mean[o] = expf(a.s_mean[o] - mx_s[h]) / den_s[h];
        a.w_sum[o] = 1.f / (1.f + expf(-a.s_sum[o]));
    }
    __syncthreads();   // the weights written above are read below by other lanes of this workgroup
    for (int c = tid; c < HD; c += 256) {
        const int h = c / a.hd;
        float am = 0.f, as = 0.f;
        for (int64_t p = p0; p < p1; ++p) {
            const size_t v = (size_t)a.perm[p];
            am = fmaf(a.w_mean[v * nh + h], a.v_mean[v * HD + c], am);
            as = fmaf(a.w_sum[v * nh + h], a.v_sum[v * HD + c], as);
        }
        a.g_mean[(size_t)g * HD + c] = am;
        a.g_sum[(size_t)g * HD + c] = as;
    }
    for (int c = tid; c < a.D; c += 256) {
        float mx = -INFINITY;
        int am = -1;
        for (int64_t p = p0; p < p1; ++p) {
            const int64_t v = a.perm[p];
            const float x = a.emb[(size_t)v * a.D + c];
            if (x > mx) { mx = x; am = (int)v; }
        }
        a.g_max[(size_t)g * a.D + c] = n > 0 ? mx : 0.f;
        a.argmax[(size_t)g * a.D + c] = am;
    }
}

// one workgroup per node; wave w takes heads w, w + 4, ... (one wave reduction per head and dot product)
//   d v_mean[v, h, :] = w_mean[v, h] dg_mean[g, h, :]
//   d s_mean[v, h]    = w_mean[v, h] <dg_mean[g, h, :], v_mean[v, h, :] - g_mean[g, h, :]>     (softmax Jacobian, sum_u w_u = 1)
//   d v_sum[v, h, :]  = w_sum[v, h] dg_sum[g, h, :]
//   d s_sum[v, h]     = w_sum (1 - w_sum) <dg_sum[g, h, :], v_sum[v, h, :]>
//   d emb[v, c]       = dg_max[g, c] if v is the arg-max of column c of its graph, else 0
__global__ __launch_bounds__(256) void k_readout_bwd(ReadoutArgs a) {
    const int v = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nh = a.nh, hd = a.hd, HD = nh * hd;
    const size_t g = (size_t)a.n2g[v];
    for (int h = wv; h < nh; h += 4) {
        const float wm = a.w_mean[(size_t)v * nh + h], ws = a.w_sum[(size_t)v * nh + h];
        float dm = 0.f, ds = 0.f;
        for (int k = lane; k < hd; k += 64) {
            const int c = h * hd + k;
            const float dgm = a.dg_mean[g * HD + c], dgs = a.dg_sum[g * HD + c];
            const float vm = a.v_mean[(size_t)v * HD + c], vs = a.v_sum[(size_t)v * HD + c];
            a.d_v_mean[(size_t)v * HD + c] = wm * dgm;
            a.d_v_sum[(size_t)v * HD + c] = ws * dgs;
            dm = fmaf(dgm, vm - a.g_mean[g * HD + c], dm);
            ds = fmaf(dgs, vs, ds);
        }
        dm = wave_sum(dm);
        ds = wave_sum(ds);
        if (lane == 0) {
            a.d_s_mean[(size_t)v * nh + h] = wm * dm;
            a.d_s_sum[(size_t)v * nh + h] = ws * (1.f - ws) * ds;
        }
    }
    for (int c = tid; c < a.D; c += 256)
        a.d_emb[(size_t)v * a.D + c] = a.argmax[g * a.D + c] == v ? a.dg_max[g * a.D + c] : 0.f;
}

// ---------------------------------------------------------------------------------------------------------------------
// Pooling BEFORE the value projection.  The value MLP of a weighted head is Linear(ReLU(Linear(x))) (graph_readout.py:219-223, one
// hidden layer of nh hd units), and what is pooled is linear in its LAST layer:
//     g[g, h, :] = sum_v w[v, h] (W2 r_v + b2)[h, :] = W2[h] (sum_v w[v, h] r_v) + b2[h] sum_v w[v, h],       r_v = ReLU(W1 x_v + b1)
// so the [V, K] x [K, nh hd] product over all V nodes (and its two backward products) becomes nh products of [G, K] x [K, hd] over
// the G graphs - at the default sizes 122 GFLOP forward + backward per read-out head against 5 - once the hidden states are pooled
// per head:   p[h, g, :] = sum_v w[v, h] r_v   ([nh, G, K]; K = 768),   wtot[g, h] = sum_v w[v, h].
// One workgroup per graph, forward and backward; every sum runs over the graph's node list in its fixed order, every output element is
// written by one thread (no atomics).  HB heads are carried per pass over the node list (12 = the default head count in one pass).
struct ReadoutHArgs {
    const float *s_mean, *s_sum;          // [V, nh] scores
    const float *h_mean, *h_sum;          // rows of K hidden values, row stride ldh (two column blocks of one [V, 4 K] activation)
    const float* emb;                     // [V, D]
    const int64_t *perm, *rowptr;
    float *w_mean, *w_sum;                // [V, nh] (forward: out, backward: in)
    float *p_mean, *p_sum;                // [nh, G, K]
    float *wtot_mean, *wtot_sum;          // [G, nh]: 1 (0 for an empty graph) | sum of the sigmoid weights
    float* g_max; int32_t* argmax;        // [G, D]
    const float *dp_mean, *dp_sum, *dwtot_sum, *dg_max;     // backward inputs ([nh, G, K] x 2, [G, nh], [G, D])
    float *d_s_mean, *d_s_sum;            // [V, nh]
    float *d_h_mean, *d_h_sum;            // [V, K] contiguous
    float* d_emb;                         // [V, D]
    int V, G, nh, K, ldh, D;
};

constexpr int READOUT_KJ_MAX = 4;         // K <= 1024 hidden units (KJ = ceil(K / 256) columns per thread)
constexpr int READOUT_NC = 64;            // nodes of a graph staged per pass (ids and weights in LDS)
constexpr int READOUT_NU = 4;             // nodes in flight per step of the node loop
constexpr int READOUT_MAX_D = 2048;       // node embedding width of the pooled-hidden kernels (the max pooler's running maxima live in registers)

// ids and weights of the nodes [c0, c0 + cn) of the graph's list -> LDS (weights of nodes beyond the list read as 0, their ids as the
// last node's: the node loop runs in steps of READOUT_NU without a tail)
template <int HB>
__device__ __forceinline__ void readout_stage(const ReadoutHArgs& a, int64_t p0, int c0, int cn, int hb, int (&vs_s)[READOUT_NC + READOUT_NU],
                                              float (&wm_s)[READOUT_NC + READOUT_NU][HB], float (&ws_s)[READOUT_NC + READOUT_NU][HB]) {
    const int tid = threadIdx.x;
    for (int i = tid; i < READOUT_NC + READOUT_NU; i += 256) vs_s[i] = (int)a.perm[p0 + c0 + min(i, cn - 1)];
    for (int idx = tid; idx < (READOUT_NC + READOUT_NU) * HB; idx += 256) {
        const int i = idx / HB, q = idx - i * HB;
        const size_t o = (size_t)a.perm[p0 + c0 + min(i, cn - 1)] * a.nh + hb + q;
        wm_s[i][q] = i < cn ? a.w_mean[o] : 0.f;
        ws_s[i][q] = i < cn ? a.w_sum[o] : 0.f;
    }
}

// (No thread walks a graph's node list through global memory on its own: a chain of dependent loads - perm[p], then the row it
// points to - costs a microsecond per node.  Ids, scores and weights of READOUT_NC nodes at a time are staged in LDS by all threads.)
template <int HB, int KJ>
__global__ __launch_bounds__(256) void k_readout_h_fwd(ReadoutHArgs a) {
    __shared__ float mx_s[READOUT_MAX_HEADS], den_s[READOUT_MAX_HEADS];
    __shared__ float sc_s[READOUT_NC][READOUT_MAX_HEADS];
    __shared__ int vs_s[READOUT_NC + READOUT_NU];
    __shared__ float wm_s[READOUT_NC + READOUT_NU][HB], ws_s[READOUT_NC + READOUT_NU][HB];
    const int g = blockIdx.x, tid = threadIdx.x, nh = a.nh, K = a.K;
    const int64_t p0 = a.rowptr[g], p1 = a.rowptr[g + 1];
    const int n = (int)(p1 - p0);
    const float* __restrict__ h_mean = a.h_mean;
    const float* __restrict__ h_sum = a.h_sum;
    // segment softmax: maximum, then the sum of exponentials, per head in list order
    float mx = -INFINITY, den = 0.f;
    for (int pass = 0; pass < 2; ++pass)
        for (int c0 = 0; c0 < n; c0 += READOUT_NC) {
            const int cn = min(READOUT_NC, n - c0);
            __syncthreads();
            for (int idx = tid; idx < cn * nh; idx += 256) {
                const int i = idx / nh, h = idx - i * nh;
                sc_s[i][h] = a.s_mean[(size_t)a.perm[p0 + c0 + i] * nh + h];
            }
            __syncthreads();
            if (tid < nh) {
                if (pass == 0) { for (int i = 0; i < cn; ++i) mx = fmaxf(mx, sc_s[i][tid]); }
                else { for (int i = 0; i < cn; ++i) den += expf(sc_s[i][tid] - mx); }
            }
        }
    if (tid < nh) { mx_s[tid] = mx; den_s[tid] = den; }
    __syncthreads();
    for (int idx = tid; idx < n * nh; idx += 256) {
        const int pl = idx / nh, h = idx - pl * nh;
        const size_t o = (size_t)a.perm[p0 + pl] * nh + h;
        a.w_mean[o] = expf(a.s_mean[o] - mx_s[h]) / den_s[h];
        a.w_sum[o] = 1.f / (1.f + expf(-a.s_sum[o]));
    }
    __syncthreads();   // the weights written above are read below by other lanes of this workgroup
    float gmx[(READOUT_MAX_D + 255) / 256];
    int gam[(READOUT_MAX_D + 255) / 256];
#pragma unroll
    for (int j = 0; j < (READOUT_MAX_D + 255) / 256; ++j) { gmx[j] = -INFINITY; gam[j] = -1; }
    for (int hb = 0; hb < nh; hb += HB) {
        float am[HB][KJ], as[HB][KJ], wt = 0.f;
#pragma unroll
        for (int q = 0; q < HB; ++q)
#pragma unroll
            for (int j = 0; j < KJ; ++j) { am[q][j] = 0.f; as[q][j] = 0.f; }
        for (int c0 = 0; c0 < n; c0 += READOUT_NC) {
            const int cn = min(READOUT_NC, n - c0);
            __syncthreads();
            readout_stage<HB>(a, p0, c0, cn, hb, vs_s, wm_s, ws_s);
            __syncthreads();
            if (tid < HB) { for (int i = 0; i < cn; ++i) wt += ws_s[i][tid]; }
            if (hb == 0) {   // the max pooler rides the first pass over the node list: READOUT_NU rows of the embedding in flight per step
                constexpr int DJ = (READOUT_MAX_D + 255) / 256;
                const int dj = (a.D + 255) / 256;
                for (int i = 0; i < cn; i += READOUT_NU) {
                    float x[READOUT_NU][DJ];
#pragma unroll
                    for (int u = 0; u < READOUT_NU; ++u)
#pragma unroll
                        for (int j = 0; j < DJ; ++j)
                            if (j < dj) x[u][j] = a.emb[(size_t)vs_s[i + u] * a.D + min(tid + 256 * j, a.D - 1)];
#pragma unroll
                    for (int u = 0; u < READOUT_NU; ++u)      // (ids beyond the list repeat the last node: never a strict maximum)
#pragma unroll
                        for (int j = 0; j < DJ; ++j)
                            if (j < dj && x[u][j] > gmx[j]) { gmx[j] = x[u][j]; gam[j] = vs_s[i + u]; }
                }
            }
            for (int i = 0; i < cn; i += READOUT_NU) {
                float hm[READOUT_NU][KJ], hs[READOUT_NU][KJ];
#pragma unroll
                for (int u = 0; u < READOUT_NU; ++u) {
                    const size_t v = (size_t)vs_s[i + u];
#pragma unroll
                    for (int j = 0; j < KJ; ++j) {
                        const int c = min(tid + 256 * j, K - 1);
                        hm[u][j] = h_mean[v * a.ldh + c];
                        hs[u][j] = h_sum[v * a.ldh + c];
                    }
                }
#pragma unroll
                for (int u = 0; u < READOUT_NU; ++u)
#pragma unroll
                    for (int q = 0; q < HB; ++q) {
                        const float wm = wm_s[i + u][q], ws = ws_s[i + u][q];
#pragma unroll
                        for (int j = 0; j < KJ; ++j) { am[q][j] = fmaf(wm, hm[u][j], am[q][j]); as[q][j] = fmaf(ws, hs[u][j], as[q][j]); }
                    }
            }
        }
        if (tid < HB) {
            a.wtot_sum[(size_t)g * nh + hb + tid] = wt;
            a.wtot_mean[(size_t)g * nh + hb + tid] = n > 0 ? 1.f : 0.f;
        }
#pragma unroll
        for (int q = 0; q < HB; ++q)
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                const int c = tid + 256 * j;
                if (c < K) {
                    const size_t o = ((size_t)(hb + q) * a.G + g) * K + c;
                    a.p_mean[o] = am[q][j]; a.p_sum[o] = as[q][j];
                }
            }
    }
#pragma unroll
    for (int j = 0; j < (READOUT_MAX_D + 255) / 256; ++j) {
        const int c = tid + 256 * j;
        if (c < a.D) {
            a.g_max[(size_t)g * a.D + c] = n > 0 ? gmx[j] : 0.f;
            a.argmax[(size_t)g * a.D + c] = gam[j];
        }
    }
}

//   d r_v (mean head) = sum_h w_mean[v, h] dp_mean[h, g, :]          dw_mean[v, h] = <r_v, dp_mean[h, g, :]>
//   d s_mean[v, h] = w_mean[v, h] (dw_mean[v, h] - sum_u w_mean[u, h] dw_mean[u, h])                       (softmax Jacobian)
//   d s_sum[v, h]  = w_sum (1 - w_sum) (dw_sum[v, h] + dwtot_sum[g, h])
//   d emb[v, c]    = dg_max[g, c] if v is the arg-max of column c of its graph, else 0
template <int HB, int KJ>
__global__ __launch_bounds__(256) void k_readout_h_bwd(ReadoutHArgs a) {
    __shared__ float red[4][READOUT_NU][2 * HB];
    __shared__ float dwm_s[READOUT_NC + READOUT_NU][HB];
    __shared__ float pd_s[READOUT_MAX_HEADS];
    __shared__ int vs_s[READOUT_NC + READOUT_NU];
    __shared__ float wm_s[READOUT_NC + READOUT_NU][HB], ws_s[READOUT_NC + READOUT_NU][HB];
    const int g = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, nh = a.nh, K = a.K;
    const int64_t p0 = a.rowptr[g], p1 = a.rowptr[g + 1];
    const int n = (int)(p1 - p0);
    const float* __restrict__ h_mean = a.h_mean;
    const float* __restrict__ h_sum = a.h_sum;
    float* __restrict__ d_h_mean = a.d_h_mean;
    float* __restrict__ d_h_sum = a.d_h_sum;
    for (int hb = 0; hb < nh; hb += HB) {
        float dm[HB][KJ], ds[HB][KJ], pd = 0.f;
#pragma unroll
        for (int q = 0; q < HB; ++q)
#pragma unroll
            for (int j = 0; j < KJ; ++j) {
                const int c = tid + 256 * j;
                const size_t o = ((size_t)(hb + q) * a.G + g) * K + min(c, K - 1);
                dm[q][j] = c < K ? a.dp_mean[o] : 0.f;      // (columns beyond K contribute nothing to the dot products)
                ds[q][j] = c < K ? a.dp_sum[o] : 0.f;
            }
        for (int c0 = 0; c0 < n; c0 += READOUT_NC) {
            const int cn = min(READOUT_NC, n - c0);
            __syncthreads();
            readout_stage<HB>(a, p0, c0, cn, hb, vs_s, wm_s, ws_s);
            __syncthreads();
            if (hb == 0) {   // the max pooler's gradient rides the first pass over the node list
                for (int c = tid; c < a.D; c += 256) {
                    const int am = a.argmax[(size_t)g * a.D + c];
                    const float dg = a.dg_max[(size_t)g * a.D + c];
                    for (int i = 0; i < cn; ++i) a.d_emb[(size_t)vs_s[i] * a.D + c] = am == vs_s[i] ? dg : 0.f;
                }
            }
            for (int i = 0; i < cn; i += READOUT_NU) {
                float hm[READOUT_NU][KJ], hs[READOUT_NU][KJ];
#pragma unroll
                for (int u = 0; u < READOUT_NU; ++u) {
                    const size_t v = (size_t)vs_s[i + u];
#pragma unroll
                    for (int j = 0; j < KJ; ++j) {
                        const int c = min(tid + 256 * j, K - 1);
                        hm[u][j] = h_mean[v * a.ldh + c];
                        hs[u][j] = h_sum[v * a.ldh + c];
                    }
                }
#pragma unroll
                for (int u = 0; u < READOUT_NU; ++u) {
                    if (i + u < cn) {
                        const size_t v = (size_t)vs_s[i + u];
#pragma unroll
                        for (int j = 0; j < KJ; ++j) {
                            const int c = tid + 256 * j;
                            float tm = 0.f, ts = 0.f;
#pragma unroll
                            for (int q = 0; q < HB; ++q) { tm = fmaf(wm_s[i + u][q], dm[q][j], tm); ts = fmaf(ws_s[i + u][q], ds[q][j], ts); }
                            if (c < K) {
                                if (hb == 0) { d_h_mean[v * K + c] = tm; d_h_sum[v * K + c] = ts; }
                                else { d_h_mean[v * K + c] += tm; d_h_sum[v * K + c] += ts; }   // (same thread, same element: ordered)
                            }
                        }
                    }
#pragma unroll
                    for (int q = 0; q < HB; ++q) {
                        float tm = 0.f, ts = 0.f;
#pragma unroll
                        for (int j = 0; j < KJ; ++j) { tm = fmaf(hm[u][j], dm[q][j], tm); ts = fmaf(hs[u][j], ds[q][j], ts); }
                        tm = wave_sum(tm); ts = wave_sum(ts);
                        if (lane == 0) { red[wv][u][q] = tm; red[wv][u][HB + q] = ts; }
                    }
                }
                __syncthreads();
                if (tid < READOUT_NU * 2 * HB) {
                    const int u = tid / (2 * HB), q = tid - u * 2 * HB;
                    if (i + u < cn) {
                        const float t = (red[0][u][q] + red[1][u][q]) + (red[2][u][q] + red[3][u][q]);
                        const size_t v = (size_t)vs_s[i + u];
                        // the raw weight gradients are parked in the outputs (and, for the softmax Jacobian, in LDS) and finished below
                        if (q < HB) { a.d_s_mean[v * nh + hb + q] = t; dwm_s[i + u][q] = t; }
                        else a.d_s_sum[v * nh + hb + q - HB] = t;
                    }
                }
                __syncthreads();
            }
            if (tid < HB) { for (int i = 0; i < cn; ++i) pd = fmaf(wm_s[i][tid], dwm_s[i][tid], pd); }
        }
        if (tid < HB) pd_s[hb + tid] = pd;
    }
    __syncthreads();
    for (int idx = tid; idx < n * nh; idx += 256) {
        const int pl = idx / nh, h = idx - pl * nh;
        const size_t o = (size_t)a.perm[p0 + pl] * nh + h;
        const float wm = a.w_mean[o], ws = a.w_sum[o];
        a.d_s_mean[o] = wm * (a.d_s_mean[o] - pd_s[h]);
        a.d_s_sum[o] = ws * (1.f - ws) * (a.d_s_sum[o] + a.dwtot_sum[(size_t)g * nh + h]);
    }
}

template <int HB, int KJ> inline void launch_readout_h_hk(const ReadoutHArgs& a, bool backward, hipStream_t st) {
    if (backward) k_readout_h_bwd<HB, KJ><<<a.G, 256, 0, st>>>(a); else k_readout_h_fwd<HB, KJ><<<a.G, 256, 0, st>>>(a);
}
template <int HB> inline void launch_readout_h_hb(const ReadoutHArgs& a, bool backward, hipStream_t st) {
    const int kj = (a.K + 255) / 256;
    if (kj <= 1) launch_readout_h_hk<HB, 1>(a, backward, st);
    else if (kj == 2) launch_readout_h_hk<HB, 2>(a, backward, st);
    else if (kj == 3) launch_readout_h_hk<HB, 3>(a, backward, st);
    else launch_readout_h_hk<HB, 4>(a, backward, st);
}
inline void launch_readout_h(const ReadoutHArgs& a, bool backward, hipStream_t st) {
    if (a.nh % 12 == 0) launch_readout_h_hb<12>(a, backward, st);
    else if (a.nh % 4 == 0) launch_readout_h_hb<4>(a, backward, st);
    else launch_readout_h_hb<1>(a, backward, st);
}

}  // namespace adkf
