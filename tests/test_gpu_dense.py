"""GPU: the dense layer on the BF16 matrix pipe (csrc/dense_x3.h through adkf_dense_forward) against float64 and against F.linear."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    return torch.device("cuda:0")


@pytest.mark.parametrize("M,K,N,bias", [(4096, 512, 64, True), (5000, 1408, 3072, True), (4099, 1024, 384, False), (8192, 1024, 128, True)])
def test_linear_matches_float64_at_least_as_well_as_the_library_gemm(dev, M, K, N, bias, monkeypatch):
    from adkf_ift_amd import dense

    monkeypatch.setattr(dense, "MIN_N", 64)    # (the kernel takes any width; linear() only sends it the wide ones, where it pays)
    monkeypatch.setattr(dense, "WEIGHT_GRAD", True)   # (off by default: no faster than the library product)

    g = torch.Generator(device="cpu").manual_seed(M + K + N)
    x = (torch.randn(M, K, generator=g) * 1.5).to(dev).requires_grad_(True)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).requires_grad_(True)
    b = torch.randn(N, generator=g).to(dev).requires_grad_(True) if bias else None
    gy = torch.randn(M, N, generator=g).to(dev)
    assert dense.takes_hip_kernel(x, w)
    y = dense.linear(x, w, b)
    gx, gw, *gb = torch.autograd.grad(y, [x, w] + ([b] if bias else []), gy)
    y0 = F.linear(x, w, b)
    gx0, gw0, *gb0 = torch.autograd.grad(y0, [x, w] + ([b] if bias else []), gy)
    xd, wd, gyd = x.detach().double(), w.detach().double(), gy.double()
    yr = xd @ wd.t() + (b.detach().double() if bias else 0.0)
    gxr, gwr = gyd @ wd, gyd.t() @ xd
    # error against float64, in units of sum_k |a_k||b_k| (what an FP32 inner product's rounding scales with)
    sy = xd.abs() @ wd.abs().t() + 1e-30
    sx = gyd.abs() @ wd.abs() + 1e-30
    e_y, e_y0 = ((y.double() - yr).abs() / sy).max().item(), ((y0.double() - yr).abs() / sy).max().item()
    e_x, e_x0 = ((gx.double() - gxr).abs() / sx).max().item(), ((gx0.double() - gxr).abs() / sx).max().item()
    assert e_y <= 5e-7 and e_x <= 5e-7, (e_y, e_x)                 # FP32 accuracy (the library product itself reaches 3e-7 on these shapes) ...
    assert e_y <= 1.5 * e_y0 + 1e-8 and e_x <= 1.5 * e_x0 + 1e-8, (e_y, e_y0, e_x, e_x0)   # ... at least the library's
    sw = gyd.abs().t() @ xd.abs() + 1e-30                       # the weight gradient: a contraction over all M rows, in row ranges
    e_w, e_w0 = ((gw.double() - gwr).abs() / sw).max().item(), ((gw0.double() - gwr).abs() / sw).max().item()
    assert e_w <= 1e-6 and e_w <= 1.5 * e_w0 + 1e-8, (e_w, e_w0)
    if bias:
        assert torch.allclose(gb[0], gb0[0], rtol=1e-5, atol=1e-4)


def test_weight_gradient_is_reproducible_and_handles_ragged_sizes(dev):
    from adkf_ift_amd import dense

    g = torch.Generator(device="cpu").manual_seed(5)
    for M, N, K in ((5003, 200, 136), (4096, 128, 1024), (9999, 33, 70)):
        gy, x = torch.randn(M, N, generator=g).to(dev), torch.randn(M, K, generator=g).to(dev)
        a, b = dense._weight_grad(gy, x), dense._weight_grad(gy, x)
        assert torch.equal(a, b)
        ref = gy.double().t() @ x.double()
        assert ((a.double() - ref).abs().max() / ref.abs().max()).item() < 2e-6, (M, N, K)


@pytest.mark.parametrize("K,N,bias", [(256, 256, False), (256, 200, True), (128, 202, True), (64, 130, False)])
def test_short_contraction_over_many_rows_takes_the_persistent_form(dev, K, N, bias):
    """adkf_dense_forward with K in {64, 128, 256} and at least one row tile per CU runs k_dense3_sk (a row tile's whole K extent in
    registers, tiles walked by a persistent workgroup, 16-byte stores of the transposed MFMA blocks; N = 202: the 4-byte store path):
    FP32 accuracy against float64, ragged last row tile and last column tile, and the same bits as the tile-per-workgroup kernel
    (which a batch of few rows still takes) on the rows both compute."""
    from adkf_ift_amd import dense

    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    M = cus * 128 + 77
    g = torch.Generator(device="cpu").manual_seed(K + N)
    x = (torch.randn(M, K, generator=g) * 1.5).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev) if bias else None
    planes = dense._split(w)
    y = dense._dense(x, planes, b, N)
    ref = x.double() @ w.double().t() + (b.double() if bias else 0.0)
    scale = x.double().abs() @ w.double().abs().t() + 1e-30
    assert (((y.double() - ref).abs() / scale).max().item()) <= 5e-7
    few = dense._dense(x[:1000].contiguous(), planes, b, N)     # 8 row tiles: the tile-per-workgroup kernel
    assert torch.equal(few, y[:1000])
    assert torch.equal(dense._dense(x, planes, b, N), y)


def test_small_and_odd_shapes_take_the_library(dev):
    from adkf_ift_amd import dense

    x = torch.randn(100, 512, device=dev)
    assert not dense.takes_hip_kernel(x, torch.randn(64, 512, device=dev))                              # few rows
    assert not dense.takes_hip_kernel(torch.randn(8192, 100, device=dev), torch.randn(64, 100, device=dev))   # K not a multiple of 32
    y = dense.linear(x, torch.ones(64, 512, device=dev))
    assert y.shape == (100, 64)


def test_non_contiguous_rows_are_handled(dev, monkeypatch):
    from adkf_ift_amd import dense

    monkeypatch.setattr(dense, "MIN_N", 64)

    big = torch.randn(6000, 2048, device=dev)
    x = big[:, 512:1536]                       # row stride 2048, 16-byte aligned start
    w = torch.randn(128, 1024, device=dev) / 32
    assert dense.takes_hip_kernel(x, w)
    y = dense.linear(x, w)
    ref = (x.double() @ w.double().t())
    assert ((y.double() - ref).abs().max() / ref.abs().max()).item() < 1e-6


def test_stand_in_feature_map_weight_gradient_both_ways(dev, monkeypatch):
    """adkf_ift_amd.synthetic.LinearFeatureMap's dW = X^T dZ at the C2 shape: the row-range kernel against the chunked bmm + sum."""
    from adkf_ift_amd import synthetic

    g = torch.Generator().manual_seed(3)
    X = torch.randn(2, 64, 128, 256, generator=g).to(dev)
    W = (torch.randn(256, 256, generator=g) / 16).to(dev).requires_grad_(True)
    gz = torch.randn(2, 64, 128, 256, generator=g).to(dev)
    out = []
    for flag in (True, False):
        monkeypatch.setattr(synthetic, "_X3_DW", flag)
        fm = synthetic.LinearFeatureMap(X[0], X[1], W)
        (dW,) = torch.autograd.grad(fm(), [W], gz)
        out.append(dW)
    ref = (X.reshape(-1, 256).double() / 16.0).t() @ gz.reshape(-1, 256).double()
    e = [((o.double() - ref).abs().max() / ref.abs().max()).item() for o in out]
    assert e[0] < 2e-6 and e[0] <= 1.5 * e[1] + 1e-7, e


def test_stand_in_feature_map_forward_both_ways(dev, monkeypatch):
    """LinearFeatureMap's Z = X W at the C2 shape (65 536 rows): k_dense3_sk through adkf_split_planes_t against the library GEMM, both
    against float64 in units of sum |x||w|; and the transposing split against the plain one on W^T, bit for bit."""
    from adkf_ift_amd import dense, synthetic

    g = torch.Generator().manual_seed(4)
    X = torch.randn(2, 256, 128, 256, generator=g).to(dev)
    W = (torch.randn(256, 256, generator=g) / 16).to(dev).requires_grad_(True)
    assert torch.equal(dense._split_t(W.detach()), dense._split(W.detach().t().contiguous()))
    out = []
    for flag in (True, False):
        monkeypatch.setattr(synthetic, "_X3_FWD", flag)
        fm = synthetic.LinearFeatureMap(X[0], X[1], W)
        out.append(fm().detach())
    rows = torch.arange(0, 65536, 97, device=dev)
    x2 = (X.reshape(-1, 256)[rows].double() / 16.0)
    ref, scale = x2 @ W.detach().double(), x2.abs() @ W.detach().double().abs() + 1e-30
    e = [((o.reshape(-1, 256)[rows].double() - ref).abs() / scale).max().item() for o in out]
    assert e[0] <= 5e-7 and e[0] <= 1.5 * e[1] + 1e-8, e
