import torch, time, math
dev = torch.device('cuda:0')
T, N, d = 256, 128, 256
X = torch.randn(2, T, N, d, device=dev)
W = torch.randn(d, d, device=dev, requires_grad=True)
def bench(name, fwd, iters=20):
    for _ in range(3):
        Z = fwd(); Z.backward(torch.ones_like(Z)); W.grad = None
    torch.cuda.synchronize()
    e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
    tf = tb = 0.0
    for _ in range(iters):
        e0.record(); Z = fwd(); e1.record(); g = torch.ones_like(Z); torch.cuda.synchronize(); e1.record(); Z.backward(g); e2.record(); torch.cuda.synchronize()
        tb += e1.elapsed_time(e2); W.grad = None
    for _ in range(iters):
        e0.record(); Z = fwd(); e1.record(); torch.cuda.synchronize(); tf += e0.elapsed_time(e1)
    print(f'{name:40s} fwd {tf/iters*1e3:7.1f} us   bwd {tb/iters*1e3:7.1f} us')
c = 1 / math.sqrt(d)
X2 = X.view(-1, d)
bench('matmul 4D @ W*c', lambda: torch.matmul(X, W * c))
bench('mm 2D @ W*c', lambda: X2 @ (W * c))
bench('mm 2D @ W then *c', lambda: (X2 @ W) * c)
Wt = W.detach().t().contiguous().requires_grad_(True)
bench('F.linear(X2, W^T)', lambda: torch.nn.functional.linear(X2, (W * c).t()))
XT = X2.t().contiguous()
bench('(W^T*c) @ X^T  -> transposed out', lambda: ((W * c).t() @ XT))

class ChunkedLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, X2, W, c, chunks):
        ctx.save_for_backward(X2); ctx.c = c; ctx.chunks = chunks
        return X2 @ (W * c)
    @staticmethod
    def backward(ctx, g):
        (X2,) = ctx.saved_tensors
        ch = ctx.chunks
        R, d = X2.shape
        part = torch.bmm(X2.view(ch, R // ch, d).transpose(1, 2), g.reshape(ch, R // ch, -1))
        return None, part.sum(0) * ctx.c, None, None
for ch in (16, 32, 64, 128):
    bench(f'chunked bmm backward, {ch} chunks', lambda ch=ch: ChunkedLinear.apply(X2, W, c, ch))
