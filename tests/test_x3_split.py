"""CPU: the arithmetic csrc/gemm_x3.h rests on, emulated in numpy (no GPU, no library call).

A float32 is the exact sum of three bfloat16 values when each cut rounds to nearest; the six products kept out of nine then give an
inner product to float32 accuracy; cutting by TRUNCATION instead leaves every inner product of same-sign operands biased low
(what the first version of the kernel showed on the device: tools/x3_bench.hip, mean signed error)."""
import numpy as np


def _bf16_rn(x):
    u = np.asarray(x, dtype=np.float32).view(np.uint32).astype(np.uint64)
    r = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16
    return r.astype(np.uint32).view(np.float32)


def _bf16_trunc(x):
    return (np.asarray(x, dtype=np.float32).view(np.uint32) & np.uint32(0xFFFF0000)).view(np.float32)


def _split3(x, cut):
    x0 = cut(x)
    r = (x - x0).astype(np.float32)
    x1 = cut(r)
    x2 = (r - x1).astype(np.float32)
    return x0, x1, x2


KEPT = [(0, 0), (0, 1), (1, 0), (0, 2), (1, 1), (2, 0)]


def test_three_bfloat16_pieces_sum_to_the_float_exactly():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.standard_normal(200000) * s for s in (1e-20, 1e-3, 1.0, 37.0, 1e20)]).astype(np.float32)
    for cut in (_bf16_rn, _bf16_trunc):
        x0, x1, x2 = _split3(x, cut)
        assert np.array_equal(cut(x2), x2), "the last remainder is a bfloat16 value: nothing is rounded away"
        assert np.array_equal((x0.astype(np.float64) + x1.astype(np.float64) + x2.astype(np.float64)).astype(np.float32), x)
        assert np.array_equal(x0.astype(np.float64) + x1.astype(np.float64) + x2.astype(np.float64), x.astype(np.float64))


def test_six_products_give_a_float32_accurate_inner_product():
    rng = np.random.default_rng(1)
    X = (rng.standard_normal((96, 256)) * 3).astype(np.float32)
    Y = (rng.standard_normal((80, 256)) * 3).astype(np.float32)
    exact = X.astype(np.float64) @ Y.astype(np.float64).T
    scale = np.abs(X).astype(np.float64) @ np.abs(Y).astype(np.float64).T
    xs, ys = _split3(X, _bf16_rn), _split3(Y, _bf16_rn)
    six = sum(xs[i].astype(np.float64) @ ys[j].astype(np.float64).T for i, j in KEPT)
    three = sum(xs[i].astype(np.float64) @ ys[j].astype(np.float64).T for i, j in KEPT[:3])
    assert np.max(np.abs(six - exact) / scale) < 2e-8       # the three dropped terms: < 2^-24 of sum |x||y| (observed 3e-9)
    assert np.max(np.abs(three - exact) / scale) > 5e-7     # ... and three products would NOT do (2^-16)
    f32 = (X @ Y.T).astype(np.float64)
    assert np.max(np.abs(six - exact) / scale) < np.max(np.abs(f32 - exact) / scale)


def test_cutting_by_truncation_biases_inner_products_and_rounding_does_not():
    rng = np.random.default_rng(2)
    X = (1.0 + rng.random((64, 64))).astype(np.float32)       # same-sign operands: the dropped terms cannot cancel
    Y = (1.0 + rng.random((64, 64))).astype(np.float32)
    exact = X.astype(np.float64) @ Y.astype(np.float64).T
    bias = {}
    for name, cut in (("rn", _bf16_rn), ("trunc", _bf16_trunc)):
        xs, ys = _split3(X, cut), _split3(Y, cut)
        six = sum(xs[i].astype(np.float64) @ ys[j].astype(np.float64).T for i, j in KEPT)
        bias[name] = float(np.mean((six - exact) / exact))
    assert bias["trunc"] < -1e-8                    # every product short by ~2^-25 (measured on the device: 6e-8 on the distances)
    assert abs(bias["rn"]) < 2e-10
