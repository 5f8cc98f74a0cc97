"""CPU: adkf_ift_amd.dense.linear is F.linear wherever the HIP kernel does not apply (CPU tensors, small or odd shapes, other dtypes)."""
import torch
import torch.nn.functional as F


def test_linear_is_f_linear_off_the_gpu_path():
    from adkf_ift_amd import dense

    g = torch.Generator().manual_seed(0)
    for M, K, N, dt in ((5000, 512, 128, torch.float32), (7, 3, 5, torch.float64), (4096, 1024, 256, torch.float64)):
        x = torch.randn(M, K, generator=g, dtype=dt, requires_grad=True)
        w = torch.randn(N, K, generator=g, dtype=dt, requires_grad=True)
        b = torch.randn(N, generator=g, dtype=dt)
        assert not dense.takes_hip_kernel(x, w)            # CPU tensors never do
        y, y0 = dense.linear(x, w, b), F.linear(x, w, b)
        assert torch.equal(y, y0)
        gx, gw = torch.autograd.grad(y.sum(), [x, w])
        gx0, gw0 = torch.autograd.grad(y0.sum(), [x, w])
        assert torch.equal(gx, gx0) and torch.equal(gw, gw0)


def test_selection_rule():
    from adkf_ift_amd import dense

    class Fake:   # shape / dtype / device facts only
        def __init__(self, shape, cuda=True, dtype=torch.float32):
            self.shape, self.is_cuda, self.dtype = shape, cuda, dtype
        def dim(self):
            return len(self.shape)

    ok = lambda M, K, N, **kw: dense.takes_hip_kernel(Fake((M, K), **kw), Fake((N, K)))
    assert ok(56554, 1408, 3072) and ok(56554, 1024, 384) and ok(56554, 1024, 128)
    assert not ok(56554, 128, 1024)          # short contraction
    assert not ok(2304, 2560, 2048)          # few rows (the fc head: one row per molecule)
    assert not ok(56554, 1000, 384)          # contraction not a multiple of 32
    assert not ok(56554, 1024, 384, cuda=False) and not ok(56554, 1024, 384, dtype=torch.float64)
