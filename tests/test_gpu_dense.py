"""GPU: adkf_dense_forward / adkf_dense_backward (csrc/dense.h) against float64 torch on the shapes the feature extractor
uses (BOOM 128 <-> 1024 on ~10^4 nodes, the 2560 -> 2048 fc head, tiny read-out layers, ragged sizes that are not multiples of
the 64-wide tile or of 4)."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,K,N,act,bias", [(5000, 128, 1024, 2, True), (5000, 1024, 128, 0, True), (2304, 2560, 2048, 1, True),
                                           (37, 1408, 768, 1, True), (130, 33, 7, 0, False), (1, 128, 12, 2, True),
                                           (3000, 3072, 384, 0, False)])
def test_dense_matches_float64(M, K, N, act, bias):
    from adkf_ift_amd import dense

    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g)
    w = torch.randn(N, K, generator=g) / K ** 0.5
    b = torch.randn(N, generator=g) if bias else None
    dy = torch.randn(M, N, generator=g)
    xs = [t.double().requires_grad_(True) for t in (x, w)] + ([b.double().requires_grad_(True)] if bias else [])
    ref = dense._torch_act(F.linear(xs[0], xs[1], xs[2] if bias else None), act)
    ref.backward(dy.double())
    xg = [t.to(dev).requires_grad_(True) for t in (x, w)] + ([b.to(dev).requires_grad_(True)] if bias else [])
    got = dense.linear(xg[0], xg[1], xg[2] if bias else None, act)
    got.backward(dy.to(dev))
    scale = lambda t: max(t.abs().max().item(), 1e-30)
    assert (got.double().cpu() - ref).abs().max().item() <= 2e-6 * scale(ref) * max(1.0, (K / 256) ** 0.5)
    for a, r in zip(xg, xs):
        assert (a.grad.double().cpu() - r.grad).abs().max().item() <= 5e-6 * scale(r.grad) * max(1.0, (max(M, N) / 256) ** 0.5), a.shape
