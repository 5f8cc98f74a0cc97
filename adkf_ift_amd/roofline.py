"""Frozen work model of one meta-task and the MI355X peaks it is priced against
(/opt/skills/guides/MI355X_MICROARCH.md: FP32 matrix = FP32 vector = 157.3 TFLOP/s spec; HBM3E 8 TB/s spec)."""
from __future__ import annotations

PEAK_FP32_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def flops_per_task(N: int, Nq: int, d: int, inner_evals: int) -> dict:
    """Algorithmic FLOPs (mul+add = 2) of one meta-task, term by term; O(N^2) elementwise work excluded."""
    pair = N * N + N * Nq + Nq * Nq
    terms = {
        "distances": 2 * d * pair,
        "inner_fit": inner_evals * N ** 3,                 # Cholesky N^3/3 + inverse 2N^3/3 per evaluation
        "hessian": 4 * N ** 3,
        "outer_nll": 3 * (N ** 3 / 3 + 2 * Nq * N * N + 2 * Nq * Nq * N + Nq ** 3 / 3),
        "mixed": 4 * N ** 3,
        "dZ": 4 * d * pair,
    }
    terms["total"] = sum(terms.values())
    return terms


def bytes_per_task(N: int, Nq: int, d: int) -> int:
    """Compulsory HBM traffic: read Z_s, Z_q, write dZ_s, dZ_q."""
    return 4 * 2 * (N + Nq) * d


def dense_flops_c3(V: int, E: int, G: int, hidden: int = 128, towers: int = 4, msg: int = 64, boom: int = 1024, layers: int = 10,
                   node_feat: int = 32, ro_heads: int = 12, ro_head_dim: int = 64, ro_out: int = 512, ecfp: int = 2048,
                   fc_hidden: int = 2048, fc_out: int = 2048) -> dict:
    """Dense-layer FLOPs (mul+add = 2) of ONE forward of the default deep-kernel model over a meta-batch of V
    nodes, E message edges (all edge types, both directions) and G molecules, as this build evaluates it: message functions per edge
    (gathered GEMM), the output projection on the UNSCALED aggregates ([V, 4 towers x 4 x 64] x [1024, 3 x 128], the three PNA scalers
    folded in afterwards), BOOM, the read-out with the value layers applied AFTER pooling (G rows instead of V).  A meta-step
    is priced at 3 x forward (dX and dW products of the backward).  Element-wise, gather / scatter and normalisation work is excluded -
    like O(N^2) work in flops_per_task."""
    in_dim = hidden // towers
    ro_hid = ro_heads * ro_head_dim
    node_dim = (layers + 1) * hidden
    t = {
        "init_proj": 2 * V * node_feat * hidden,
        "messages": layers * 2 * E * towers * (2 * in_dim) * (3 * msg),
        "msg_out_projection": layers * 2 * V * (towers * 4 * msg) * (3 * hidden),
        "boom": layers * 2 * 2 * V * hidden * boom,
        "readout_first": 2 * V * node_dim * 4 * ro_hid,
        "readout_scores": 2 * 2 * V * ro_hid * ro_heads,
        "readout_values_pooled": 2 * 2 * G * ro_hid * ro_hid,
        "readout_combine": 2 * G * (2 * ro_hid * ro_out + node_dim * ro_out + 3 * ro_out * ro_out),
        "fc_head": 2 * G * ((ro_out + ecfp) * fc_hidden + fc_hidden * fc_out),
    }
    t["forward"] = sum(t.values())
    t["step"] = 3 * t["forward"]
    return t
