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
};

__global__ __launch_bounds__(256) void k_pna_fwd(PnaArgs a) {
    const int v = blockIdx.x, H = a.H, m = a.m;
    const int64_t p0 = a.rowptr[v], p1 = a.rowptr[v + 1];
    const float deg = (float)(p1 - p0);
    for (int idx = threadIdx.x; idx < H * m; idx += blockDim.x) {
        const int h = idx / m, f = idx - h * m;
        float s = 0.f, b = 0.f, mx = -INFINITY;
        int am = -1;
        for (int64_t p = p0; p < p1; ++p) {
            const int64_t e = a.perm[p];
            const float* row = a.msgs + ((size_t)e * H + h) * 3 * m;
            s += row[f];
            b += row[m + f];
            const float c = row[2 * m + f];
            if (c > mx) { mx = c; am = (int)e; }
        }
        const float mean = p1 > p0 ? b / deg : 0.f;
        float dev = 0.f;
        for (int64_t p = p0; p < p1; ++p) {
            const float bb = a.msgs[((size_t)a.perm[p] * H + h) * 3 * m + m + f];
            dev += fmaxf(bb * bb - mean * mean, 0.f) + PNA_SMALL;
        }
        float* out = a.agg + ((size_t)v * H + h) * 4 * m;
        out[f] = s;
        out[m + f] = mean;
        out[2 * m + f] = sqrtf(dev);
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
        const float mean = ag[m + f], sd = ag[2 * m + f];
        const float d_sum = dg[f], d_mean = dg[m + f], d_std = dg[2 * m + f], d_max = dg[3 * m + f];
        const float g = sd > 0.f ? d_std / (2.f * sd) : 0.f;
        const int am = a.argmax[((size_t)v * H + h) * m + f];
        float cnt = 0.f;
        for (int64_t p = p0; p < p1; ++p) {
            const float bb = a.msgs[((size_t)a.perm[p] * H + h) * 3 * m + m + f];
            cnt += (bb * bb > mean * mean) ? 1.f : 0.f;
        }
        const float via_mean = (d_mean - 2.f * mean * g * cnt) / deg;
        for (int64_t p = p0; p < p1; ++p) {
            const int64_t e = a.perm[p];
            const size_t o = ((size_t)e * H + h) * 3 * m;
            const float bb = a.msgs[o + m + f];
            a.d_msgs[o + f] = d_sum;
            a.d_msgs[o + m + f] = ((bb * bb > mean * mean) ? 2.f * bb * g : 0.f) + via_mean;
            a.d_msgs[o + 2 * m + f] = ((int)e == am) ? d_max : 0.f;
        }
    }
}

}  // namespace adkf
