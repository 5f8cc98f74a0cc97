// The element-wise middle of a GNN block (fs_mol/modules/gnn.py:477-515, GNNBlock.forward with the PNA scalers folded behind
// the output projection as adkf_ift_amd/gnn.py::GNNBlock does) as ONE kernel forward and ONE backward:
//     new  = p0 + amplify[v] p1 + attenuate[v] p2 + bias          p = [p0 | p1 | p2]: the projected (unscaled) aggregates, [V, 3 hid]
//     x1   = x + alpha new                                         (ReZero residual, :497-503)
//     h    = LayerNorm(x1) = (x1 - mean) rstd gamma + beta         (boom_norm_layer, :505-507: the input of the BOOM MLP)
// In PyTorch these are three slice-adds, two broadcast multiplies, the bias add, the alpha multiply, the residual add and the layer
// norm forward - nine launches over [V, hid] tensors per block - and about twenty in the backward, five of them column or full
// reductions over V (d bias, d gamma, d beta, d alpha).  The backward here recomputes `new` from p (cheaper than storing it),
// writes d p and d x once, and reduces the four parameter gradients DETERMINISTICALLY: every workgroup owns a fixed range of rows,
// its waves fixed rows of that range, the per-workgroup partials are combined by k_block_reduce in workgroup order.
// One wave per row, hid / 64 columns per lane (hid a multiple of 64, at most 256; other widths keep the PyTorch path).
#pragma once
#include "device_utils.h"

namespace adkf {

constexpr int BLK_ROWS = 128;      // rows per workgroup of the backward (4 waves x 32 rows): the unit of the ordered reduction
constexpr int BLK_MAXC = 4;        // hid <= 256

struct BlockArgs {
    const float *p, *x, *amp, *att, *bias, *alpha, *gamma, *beta;
    float *x1, *h, *mu, *rstd;                  // forward outputs (mu, rstd [V]: kept for the backward)
    const float *g_x1, *g_h;                    // backward inputs
    float *d_p, *d_x, *part;                    // part [nwg][3 hid + 1]: d bias | d gamma | d beta | d alpha
    float eps; int V, hid;
};

// new = ((p0 + amp p1) + att p2) + bias with every product and sum rounded on its own, in the order PyTorch evaluates the expression
// of adkf_ift_amd/gnn.py::GNNBlock (no fused multiply-add): the node states of the fused and of the unfused path are then the same
// numbers, not merely equally accurate ones
__device__ __forceinline__ float blk_new(float p0, float p1, float p2, float am, float at, float b) {
    return __fadd_rn(__fadd_rn(__fadd_rn(p0, __fmul_rn(am, p1)), __fmul_rn(at, p2)), b);
}

template <int C>   // C = hid / 64
__global__ __launch_bounds__(256) void k_block_fwd(BlockArgs a) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, hid = 64 * C;
    const float alpha = a.alpha[0];
    float bias[C], gam[C], bet[C];
#pragma unroll
    for (int c = 0; c < C; ++c) { bias[c] = a.bias[lane + 64 * c]; gam[c] = a.gamma[lane + 64 * c]; bet[c] = a.beta[lane + 64 * c]; }
    for (long v = (long)blockIdx.x * 4 + wv; v < a.V; v += (long)gridDim.x * 4) {
        const float am = a.amp[v], at = a.att[v];
        const float* pr = a.p + (size_t)v * 3 * hid;
        float x1[C], s = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int j = lane + 64 * c;
            const float nw = blk_new(pr[j], pr[hid + j], pr[2 * hid + j], am, at, bias[c]);
            x1[c] = __fadd_rn(a.x[(size_t)v * hid + j], __fmul_rn(alpha, nw));
            s += x1[c];
        }
        const float mean = wave_sum(s) / (float)hid;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) { const float dlt = x1[c] - mean; q += dlt * dlt; }
        const float rstd = rsqrtf(wave_sum(q) / (float)hid + a.eps);     // (biased variance, as torch.nn.LayerNorm)
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int j = lane + 64 * c;
            a.x1[(size_t)v * hid + j] = x1[c];
            a.h[(size_t)v * hid + j] = (x1[c] - mean) * rstd * gam[c] + bet[c];
        }
        if (lane == 0) { a.mu[v] = mean; a.rstd[v] = rstd; }
    }
}

// G = g_x1 + LayerNorm-backward(g_h) is the total gradient at x1;  d x = G,  d new = alpha G,  d p = (1 | amp | att) d new,
// d bias = sum_v d new,  d alpha = sum G . new,  d gamma = sum_v g_h xhat,  d beta = sum_v g_h.
template <int C>
__global__ __launch_bounds__(256) void k_block_bwd(BlockArgs a) {
    __shared__ float red[4][3 * 64 * BLK_MAXC + 1];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, hid = 64 * C;
    const float alpha = a.alpha[0], ihid = 1.f / (float)hid;
    float bias[C], gam[C], db[C], dg[C], dbt[C], dal = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { bias[c] = a.bias[lane + 64 * c]; gam[c] = a.gamma[lane + 64 * c]; db[c] = dg[c] = dbt[c] = 0.f; }
    const long r0 = (long)blockIdx.x * BLK_ROWS, r1 = min((long)a.V, r0 + BLK_ROWS);
    for (long v = r0 + wv; v < r1; v += 4) {
        const float am = a.amp[v], at = a.att[v], mean = a.mu[v], rstd = a.rstd[v];
        const float* pr = a.p + (size_t)v * 3 * hid;
        float nw[C], xh[C], gh[C], s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int j = lane + 64 * c;
            nw[c] = blk_new(pr[j], pr[hid + j], pr[2 * hid + j], am, at, bias[c]);
            xh[c] = (a.x1[(size_t)v * hid + j] - mean) * rstd;
            gh[c] = a.g_h[(size_t)v * hid + j];
            const float dxh = gh[c] * gam[c];
            s1 += dxh; s2 += dxh * xh[c];
        }
        s1 = wave_sum(s1) * ihid; s2 = wave_sum(s2) * ihid;
        float* dpr = a.d_p + (size_t)v * 3 * hid;
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const int j = lane + 64 * c;
            // (this stage in float64 changes nothing measurable end to end - tried; what limits the extractor's gradients is the
            // float32 rounding of the MESSAGES in front of the std aggregation: tests/test_gpu_gnn.py, tools/diag_gnn_states.py)
            const float G = a.g_x1[(size_t)v * hid + j] + rstd * (gh[c] * gam[c] - s1 - xh[c] * s2);
            const float dn = alpha * G;
            a.d_x[(size_t)v * hid + j] = G;
            dpr[j] = dn; dpr[hid + j] = am * dn; dpr[2 * hid + j] = at * dn;
            db[c] += dn; dg[c] += gh[c] * xh[c]; dbt[c] += gh[c]; dal += G * nw[c];
        }
    }
    dal = wave_sum(dal);
#pragma unroll
    for (int c = 0; c < C; ++c) { red[wv][lane + 64 * c] = db[c]; red[wv][hid + lane + 64 * c] = dg[c]; red[wv][2 * hid + lane + 64 * c] = dbt[c]; }
    if (lane == 0) red[wv][3 * hid] = dal;
    __syncthreads();
    float* out = a.part + (size_t)blockIdx.x * (3 * hid + 1);
    for (int i = threadIdx.x; i < 3 * hid + 1; i += 256) out[i] = (red[0][i] + red[1][i]) + (red[2][i] + red[3][i]);
}

// out[i] = sum over the workgroups' partials, in workgroup order (i < 3 hid + 1: d bias | d gamma | d beta | d alpha)
__global__ __launch_bounds__(64) void k_block_reduce(const float* part, int nwg, int n, float* d_bias, float* d_gamma, float* d_beta, float* d_alpha, int hid) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    int w = 0;
    for (; w + 8 <= nwg; w += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = part[(size_t)(w + u) * n + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; w < nwg; ++w) s += part[(size_t)w * n + i];
    if (i < hid) d_bias[i] = s;
    else if (i < 2 * hid) d_gamma[i - hid] = s;
    else if (i < 3 * hid) d_beta[i - 2 * hid] = s;
    else d_alpha[0] = s;
}

}  // namespace adkf
