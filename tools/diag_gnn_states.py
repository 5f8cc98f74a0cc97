#!/usr/bin/env python
"""Where does the float32 backward of the default-width extractor leave the float64 one?  Gradient arriving at every node state
(states[k] = output of block k - 1), device vs restatement: relative L2 and max error, and the node with the largest error."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, torch.nn.functional as F
from adkf_ift_amd import gnn as G
from oracle import gnn_oracle as GO
from test_gnn import random_graphs, unit_gain_reference_state_dict

dev = torch.device("cuda:0")
cfg = G.GraphFeatureExtractorConfig()
sd = {k: v.requires_grad_(True) for k, v in unit_gain_reference_state_dict(cfg, seed=2).items()}
batch = random_graphs(40, seed=11)
w = torch.randn(40, 512, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
b32 = batch.to(dev); b32.node_features = b32.node_features.float()
model = G.GraphFeatureExtractor(cfg)
model.load_reference_state_dict({k: v.detach().float() for k, v in sd.items()})
model = model.to(dev)
model.gnn.state_grads = {}
got = model(b32)
(got * w.float().to(dev)).sum().backward()
dg = {k: v.double().cpu() for k, v in model.gnn.state_grads.items()}
# float64: the restatement's own loop with the states kept
pfx = "graph_feature_extractor."
p = {k[len(pfx):]: v for k, v in sd.items()}
g, r = cfg.gnn_config, cfg.readout_config
x = GO._linear(batch.node_features, p, "init_node_proj", bias=False)
adj = [torch.cat((a, torch.flip(a, dims=(1,))), dim=0) for a in batch.adjacency_lists]
states = [x]
for b in range(g.num_layers):
    x = GO.gnn_block(x, adj, p, f"gnn.gnn_blocks.{b}", g)
    states.append(x)
for s_ in states:
    s_.retain_grad()
node_repr = torch.cat(states, dim=-1)
Gn = batch.num_graphs
mean_r = GO.weighted_readout(node_repr, batch.node_to_graph, Gn, p, "readout._weighted_mean_pooler", r, "weighted_mean")
sum_r = GO.weighted_readout(node_repr, batch.node_to_graph, Gn, p, "readout._weighted_sum_pooler", r, "weighted_sum")
max_r = GO._linear(GO.scatter_max(node_repr, batch.node_to_graph, Gn), p, "readout._max_pooler._combination_layer", bias=False)
out = GO._linear(F.relu(torch.cat((mean_r, sum_r, max_r), dim=1)), p, "readout._combination_layer", bias=False)
(out * w).sum().backward()
deg = torch.bincount(torch.cat([a[:, 1] for a in adj]), minlength=states[0].shape[0])
for k in sorted(dg, reverse=True):
    ref = states[k].grad
    e = dg[k] - ref
    v = int(e.abs().max(dim=1).values.argmax())
    print("state %2d: rel L2 err %.2e  max err / max |grad| %.2e  worst node %4d (in-degree %d, graph %d, |grad row| %.2e of max row %.2e)" % (
        k, float(e.norm() / ref.norm()), float(e.abs().max() / ref.abs().max()), v, int(deg[v]), int(batch.node_to_graph[v]),
        float(ref[v].norm()), float(ref.norm(dim=1).max())))
