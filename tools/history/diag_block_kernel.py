#!/usr/bin/env python
"""Accuracy of the fused block-combine kernel (csrc/block.h) in isolation: forward outputs and every backward output against the
same formulas in float64, next to what the float32 PyTorch ops give.  Inputs of realistic magnitude, PNA scalers of a real
degree distribution (isolated nodes included: attenuate = 1.15e7)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.nn.functional as F
from adkf_ift_amd.gnn import _BlockCombine

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
V, hid = 20000, 128
deg = torch.randint(0, 5, (V,), generator=g).float()
amp = (torch.log(deg + 1) / 1.1515)
att = (1.1515 / (torch.log(deg + 1) + 1e-7))
p = torch.randn(V, 3 * hid, generator=g)
p[deg == 0, hid:] = 0.0
x = torch.randn(V, hid, generator=g)
bias, gamma, beta = torch.randn(hid, generator=g) * 0.1, 1 + 0.1 * torch.randn(hid, generator=g), 0.1 * torch.randn(hid, generator=g)
alpha = torch.tensor([0.7])
g1, g2 = torch.randn(V, hid, generator=g), torch.randn(V, hid, generator=g)

def unfused(p, x, amp, att, bias, alpha, gamma, beta):
    new = p[:, :hid] + amp[:, None] * p[:, hid:2 * hid] + att[:, None] * p[:, 2 * hid:] + bias
    x1 = x + alpha * new
    return x1, F.layer_norm(x1, (hid,), gamma, beta)

def run(fn, dt, device):
    ins = [t.to(device=device, dtype=dt).requires_grad_(t is not amp and t is not att) for t in (p, x, amp, att, bias, alpha, gamma, beta)]
    x1, h = fn(*ins)
    (x1 * g1.to(device=device, dtype=dt)).sum().add((h * g2.to(device=device, dtype=dt)).sum()).backward()
    outs = {"x1": x1, "h": h, "d_p": ins[0].grad, "d_x": ins[1].grad, "d_bias": ins[4].grad, "d_alpha": ins[5].grad, "d_gamma": ins[6].grad, "d_beta": ins[7].grad}
    return {k: v.detach().double().cpu() for k, v in outs.items()}

ref = run(unfused, torch.float64, "cpu")
t32 = run(unfused, torch.float32, dev)
fused = run(lambda p, x, amp, att, bias, alpha, gamma, beta: _BlockCombine.apply(p, x, amp, att, bias, alpha, gamma, beta, 1e-5), torch.float32, dev)
live = deg > 0      # rows whose d_p matters (isolated nodes: the aggregates they multiply are exactly zero)
for k in ref:
    def err(o):
        a, r = o[k], ref[k]
        if k == "d_p":
            a, r = a[live], r[live]
        return ((a - r).abs().max() / r.abs().max()).item()
    print("%-8s fused %.2e   float32 PyTorch %.2e" % (k, err(fused), err(t32)))
