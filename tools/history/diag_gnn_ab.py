#!/usr/bin/env python
"""Fused vs unfused middle-of-block path on the GPU, same process, against float64: value of the worst entries and the relative L2
error of every gradient tensor (a max-norm over 25 M entries is an extreme-value statistic; the L2 error shows whether one path is
systematically less accurate)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from adkf_ift_amd import gnn as G
from oracle import gnn_oracle as GO
from test_gnn import grads_under_reference_names, random_graphs, unit_gain_reference_state_dict

dev = torch.device("cuda:0")
cfg = G.GraphFeatureExtractorConfig()
sd = {k: v.requires_grad_(True) for k, v in unit_gain_reference_state_dict(cfg, seed=2).items()}
batch = random_graphs(40, seed=11)
w = torch.randn(40, 512, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
b32 = batch.to(dev); b32.node_features = b32.node_features.float()

def device_grads(fused):
    G._FUSED_BLOCK = fused
    model = G.GraphFeatureExtractor(cfg)
    model.load_reference_state_dict({k: v.detach().float() for k, v in sd.items()})
    model = model.to(dev)
    got = model(b32)
    (got * w.float().to(dev)).sum().backward()
    return {k: v.double().cpu() for k, v in grads_under_reference_names(model).items()}

gf, gu = device_grads(True), device_grads(False)
want = GO.graph_feature_extractor(batch, sd, cfg)
(want * w).sum().backward()
ref = {k: v.grad for k, v in sd.items() if v.grad is not None}
scale = max(g.abs().max().item() for g in ref.values())
tot = {"f": 0.0, "u": 0.0, "fu": 0.0, "n": 0.0}
rows = []
for k, r in ref.items():
    ef, eu = gf[k] - r, gu[k] - r
    tot["f"] += float((ef ** 2).sum()); tot["u"] += float((eu ** 2).sum()); tot["fu"] += float(((gf[k] - gu[k]) ** 2).sum()); tot["n"] += float((r ** 2).sum())
    rows.append((ef.abs().max().item() / scale, eu.abs().max().item() / scale, k, int(ef.abs().argmax())))
rows.sort(reverse=True)
print("relative L2 error over ALL gradient entries: fused %.3e  unfused %.3e  |fused - unfused| %.3e" % ((tot["f"] / tot["n"]) ** 0.5, (tot["u"] / tot["n"]) ** 0.5, (tot["fu"] / tot["n"]) ** 0.5))
for ef, eu, k, i in rows[:6]:
    r, a, b = ref[k].reshape(-1)[i].item(), gf[k].reshape(-1)[i].item(), gu[k].reshape(-1)[i].item()
    print("max err fused %.2e unfused %.2e  %-75s entry %d: float64 %.9g fused %.9g unfused %.9g" % (ef, eu, k[24:], i, r, a, b))
