#!/usr/bin/env python
"""Per-parameter gradient error of the default-width extractor on the GPU against the float64 restatement (the batch of
tests/test_gpu_gnn.py::test_default_width_extractor_forward_and_gradients_vs_oracle).  ADKF_GNN_FUSED_BLOCK=0 for the A/B."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch
from adkf_ift_amd.gnn import GraphFeatureExtractor, GraphFeatureExtractorConfig
from oracle import gnn_oracle as GO
from test_gnn import grads_under_reference_names, random_graphs, unit_gain_reference_state_dict

dev = torch.device("cuda:0")
cfg = GraphFeatureExtractorConfig()
sd = {k: v.requires_grad_(True) for k, v in unit_gain_reference_state_dict(cfg, seed=2).items()}
batch = random_graphs(40, seed=11)
model = GraphFeatureExtractor(cfg)
model.load_reference_state_dict({k: v.detach().float() for k, v in sd.items()})
model = model.to(dev)
b32 = batch.to(dev); b32.node_features = b32.node_features.float()
caps, amaxes = [], []
for blk in model.gnn.gnn_blocks:
    blk.mp.capture, blk.mp.capture_argmax = caps, amaxes
got = model(b32)
w = torch.randn(got.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
(got * w.float().to(dev)).sum().backward()
mine = grads_under_reference_names(model)
forced = os.environ.get("DIAG_FORCE", "")
kw = {}
if "std" in forced: kw["std_masks"] = GO.std_indicators(caps, batch.adjacency_lists, batch.node_features.shape[0], cfg.gnn_config.per_head_dim)
if "max" in forced: kw["argmaxes"] = [a.cpu() for a in amaxes]
if "relu" in forced: kw["relu_masks"] = [(c > 0).cpu() for c in caps]
want = GO.graph_feature_extractor(batch, sd, cfg, **kw)
(want * w).sum().backward()
scale = max(v.grad.abs().max().item() for v in sd.values() if v.grad is not None)
rows = []
for k, v in sd.items():
    if v.grad is None: continue
    d = (mine[k].double().cpu() - v.grad).abs()
    rows.append((d.max().item() / scale, k, v.grad.abs().max().item() / scale, int(d.argmax())))
rows.sort(reverse=True)
print("forced kinks:", forced or "-", "| fused block kernel:", os.environ.get("ADKF_GNN_FUSED_BLOCK", "1"), " forward rel err", ((got.double().cpu() - want).abs().max() / want.abs().max()).item())
for e, k, g, i in rows[:12]:
    print("%.3e  %-90s |grad|max/scale %.2e  argmax %d" % (e, k, g, i))
