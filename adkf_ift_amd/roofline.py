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
