"""Naive restatement of the reference GNN feature extractor.  TEST INFRASTRUCTURE ONLY.

Follows the reference module by module and loop by loop, with PARAMETERS UNDER THE REFERENCE'S OWN NAMES (a plain dict of
tensors, i.e. what ``torch.load(checkpoint)["model_state_dict"]`` holds), using explicit Python loops over nodes /
graphs for every ``torch_scatter`` call so that the empty-segment conventions are spelled out:

  GraphFeatureExtractor.forward   fs_mol/modules/graph_feature_extractor.py:76-98
  GNN.forward                     fs_mol/modules/gnn.py:530-556   (bidirectional edges :540-544)
  GNNBlock.forward                fs_mol/modules/gnn.py:477-515   (mp_norm_layer is NOT applied there)
  RelationalMP.forward            fs_mol/modules/gnn.py:127-148
  RelationalMultiAggrMP._aggregate_messages   fs_mol/modules/gnn.py:197-265
  CombinedGraphReadout / MultiHeadWeightedGraphReadout / UnweightedGraphReadout   fs_mol/modules/graph_readout.py:119-296

torch_scatter itself is not installable here (parity unpinned for its kernels); its documented semantics are restated:
scatter_sum / scatter_mean / scatter_max of an EMPTY segment give 0, scatter_softmax normalises within each segment.
"""
from __future__ import annotations

from typing import Dict, List

import torch
import torch.nn.functional as F

SMALL_NUMBER = 1e-7


def _linear(x, p: Dict[str, torch.Tensor], name: str, bias: bool = True):
    y = x @ p[name + ".weight"].t()
    return y + p[name + ".bias"] if bias else y


def _mlp(x, p, name: str, n_hidden: int):
    """fs_mol/modules/mlp.py: Linear, ReLU, ..., Linear under ``<name>._layers.{0,2,...}``."""
    for l in range(n_hidden):
        x = F.relu(_linear(x, p, f"{name}._layers.{2 * l}"))
    return _linear(x, p, f"{name}._layers.{2 * n_hidden}")


def _segments(index: torch.Tensor, n: int) -> List[torch.Tensor]:
    return [torch.nonzero(index == i).flatten() for i in range(n)]


def scatter_sum(src, index, n):
    return torch.stack([src[s].sum(0) if len(s) else src.new_zeros(src.shape[1:]) for s in _segments(index, n)])


def scatter_mean(src, index, n):
    return torch.stack([src[s].mean(0) if len(s) else src.new_zeros(src.shape[1:]) for s in _segments(index, n)])


def scatter_max(src, index, n):
    return torch.stack([src[s].max(0).values if len(s) else src.new_zeros(src.shape[1:]) for s in _segments(index, n)])


def scatter_softmax(src, index, n):
    out = torch.zeros_like(src)
    for s in _segments(index, n):
        if len(s):
            out[s] = torch.softmax(src[s], dim=0)
    return out


def multi_aggr_mp(x, adj_lists, p, name: str, cfg, std_mask=None, argmax=None, relu_mask=None, msg_hook=None) -> torch.Tensor:
    """One tower: RelationalMultiAggrMP with PNA scalers (cfg.type == 'PNA').
    ``std_mask`` [E, m] bool (tests only): the indicator [b_e^2 > mean^2] of the std aggregation is TAKEN from the caller instead of
    being evaluated here.  The reference's relu(b^2 - mean^2) has a kink with slope up to 1 / (2 sqrt(1e-7)) = 1581 behind it, so
    which side a nearly-equal message falls on is decided by the rounding of its inputs; with the float32 device's own indicators
    the float64 restatement is the SMOOTH function the device evaluated, and its gradients are comparable at 1e-5 instead of 1e-3.
    ``argmax`` [V, m] int (tests only; -1: no incoming message): likewise the winner of the max aggregation - between two nearly equal
    messages the float32 and the float64 forward may pick different ones, and the gradient then flows to a different edge."""
    msgs, tgts_all, e0 = [], [], 0
    for et, adj in enumerate(adj_lists):
        srcs, tgts = adj[:, 0], adj[:, 1]
        m = _mlp(torch.cat((x[srcs], x[tgts]), dim=1), p, f"{name}.message_fns.{et}", cfg.message_function_depth - 1)
        # relu_mask [E_all, 3 m] (tests only): which messages the caller's forward let through - the same idea as std_mask for the
        # ReLU behind the message function (an element within rounding distance of 0 is on or off depending on the arithmetic)
        msgs.append(F.relu(m) if relu_mask is None else torch.where(relu_mask[e0:e0 + m.shape[0]], m, torch.zeros_like(m)))
        e0 += m.shape[0]
        tgts_all.append(tgts)
    messages, targets = torch.cat(msgs), torch.cat(tgts_all)
    if msg_hook is not None:   # tests only: the post-ReLU messages [E_all, 3 m] of this tower pass through the caller (e.g. a float32 rounding)
        messages = msg_hook(name, messages, targets)
    V, m = x.shape[0], cfg.per_head_dim
    if cfg.type.lower() == "plain":
        return scatter_sum(messages, targets, V)
    s_sum = scatter_sum(messages[:, :m], targets, V)
    mean_messages = messages[:, m:2 * m]
    s_mean = scatter_mean(mean_messages, targets, V)
    diff = mean_messages.pow(2) - s_mean[targets].pow(2)
    dev = (F.relu(diff) if std_mask is None else torch.where(std_mask, diff, torch.zeros_like(diff))) + SMALL_NUMBER
    s_std = torch.sqrt(scatter_sum(dev, targets, V))
    if argmax is None:
        s_max = scatter_max(messages[:, 2 * m:3 * m], targets, V)
    else:
        am = argmax.long()
        s_max = torch.where(am >= 0, messages[:, 2 * m:3 * m].gather(0, am.clamp(min=0)), torch.zeros((), dtype=messages.dtype))
    out = torch.cat((s_sum, s_mean, s_std, s_max), dim=1)
    if cfg.type.lower() == "pna":
        deg = scatter_sum(torch.ones_like(targets).unsqueeze(-1), targets, V).squeeze(-1)
        delta = 1.1515
        log_deg = torch.log(deg.to(x.dtype) + 1).unsqueeze(-1)
        out = torch.cat((out, (log_deg / delta) * out, (delta / (log_deg + SMALL_NUMBER)) * out), dim=1)
    return out


def gnn_block(x, adj_lists, p, name: str, cfg, std_mask=None, argmax=None, relu_mask=None, msg_hook=None) -> torch.Tensor:
    """``std_mask`` [E, heads, m], ``argmax`` [V, heads, m], ``relu_mask`` [E, heads, 3 m] (see multi_aggr_mp)."""
    in_dim = cfg.hidden_dim // cfg.num_heads
    agg = [multi_aggr_mp(x[:, h * in_dim:(h + 1) * in_dim], adj_lists, p, f"{name}.mp_layers.{h}", cfg,
                         None if std_mask is None else std_mask[:, h], None if argmax is None else argmax[:, h],
                         None if relu_mask is None else relu_mask[:, h], msg_hook) for h in range(cfg.num_heads)]
    new = _linear(torch.cat(agg, dim=-1), p, f"{name}.msg_out_projection")
    if cfg.use_rezero_scaling:
        new = p[f"{name}.alpha"] * new
    x = x + new
    if cfg.intermediate_dim > 0:
        ln = F.layer_norm(x, (cfg.hidden_dim,), p[f"{name}.boom_norm_layer.weight"], p[f"{name}.boom_norm_layer.bias"])
        b = _linear(F.leaky_relu(_linear(ln, p, f"{name}.boom_layer.linear1")), p, f"{name}.boom_layer.linear2")
        if cfg.use_rezero_scaling:
            b = p[f"{name}.alpha"] * b
        x = x + b
    return x


def weighted_readout(x, n2g, G, p, name: str, rcfg, kind: str):
    scores = _mlp(x, p, f"{name}._scoring_module", 1)
    weights = torch.sigmoid(scores) if kind == "weighted_sum" else scatter_softmax(scores, n2g, G)
    values = _mlp(x, p, f"{name}._transformation_mlp", 1).view(-1, rcfg.num_heads, rcfg.head_dim)
    per_graph = scatter_sum((weights.unsqueeze(-1) * values).reshape(x.shape[0], -1), n2g, G)
    return _linear(per_graph, p, f"{name}._combination_layer", bias=False)


def graph_feature_extractor(batch, p: Dict[str, torch.Tensor], cfg, prefix: str = "graph_feature_extractor.", std_masks=None,
                            argmaxes=None, relu_masks=None, msg_hook=None) -> torch.Tensor:
    """``std_masks`` / ``argmaxes``: one [E_all, heads, m] bool / [V, heads, m] int tensor per block (see multi_aggr_mp), or None for
    the reference's own indicators / winners.  ``msg_hook(name, messages, targets) -> messages`` (tests only): every tower's post-ReLU
    messages pass through it (``name`` = "gnn.gnn_blocks.<b>.mp_layers.<h>")."""
    p = {k[len(prefix):]: v for k, v in p.items() if k.startswith(prefix)}
    g, r = cfg.gnn_config, cfg.readout_config
    x = _linear(batch.node_features, p, "init_node_proj", bias=False)
    adj = list(batch.adjacency_lists)
    if g.make_edges_bidirectional:
        adj = [torch.cat((a, torch.flip(a, dims=(1,))), dim=0) for a in adj]
    states = [x]
    for b in range(g.num_layers):
        x = gnn_block(x, adj, p, f"gnn.gnn_blocks.{b}", g, None if std_masks is None else std_masks[b],
                      None if argmaxes is None else argmaxes[b], None if relu_masks is None else relu_masks[b], msg_hook)
        states.append(x)
    node_repr = torch.cat(states, dim=-1) if r.use_all_states else states[-1]
    G = batch.num_graphs
    mean_r = weighted_readout(node_repr, batch.node_to_graph, G, p, "readout._weighted_mean_pooler", r, "weighted_mean")
    sum_r = weighted_readout(node_repr, batch.node_to_graph, G, p, "readout._weighted_sum_pooler", r, "weighted_sum")
    max_r = _linear(scatter_max(node_repr, batch.node_to_graph, G), p, "readout._max_pooler._combination_layer", bias=False)
    return _linear(F.relu(torch.cat((mean_r, sum_r, max_r), dim=1)), p, "readout._combination_layer", bias=False)


def random_reference_state_dict(cfg, seed: int = 0, dtype=torch.float64, prefix: str = "graph_feature_extractor.") -> Dict[str, torch.Tensor]:
    """A state dict with the reference's names and shapes (what a real checkpoint would contain), random values.
    alpha is drawn O(1) instead of the 1e-7 initial value so that every layer matters in the comparison."""
    gen = torch.Generator().manual_seed(seed)
    rnd = lambda *s: torch.randn(*s, generator=gen, dtype=dtype) * 0.3
    g, r = cfg.gnn_config, cfg.readout_config
    sd = {"init_node_proj.weight": rnd(g.hidden_dim, cfg.initial_node_feature_dim)}
    in_dim = g.hidden_dim // g.num_heads
    out_msg = (1 if g.type.lower() == "plain" else 3) * g.per_head_dim
    per_tower = {"plain": 1, "multiaggr": 4, "pna": 12}[g.type.lower()] * g.per_head_dim
    for b in range(g.num_layers):
        pre = f"gnn.gnn_blocks.{b}."
        sd[pre + "alpha"] = torch.rand(1, generator=gen, dtype=dtype) + 0.5
        dims = [2 * in_dim] * g.message_function_depth + [out_msg]
        for h in range(g.num_heads):
            for et in range(g.num_edge_types):
                for l in range(g.message_function_depth):
                    sd[f"{pre}mp_layers.{h}.message_fns.{et}._layers.{2 * l}.weight"] = rnd(dims[l + 1], dims[l])
                    sd[f"{pre}mp_layers.{h}.message_fns.{et}._layers.{2 * l}.bias"] = rnd(dims[l + 1])
        sd[pre + "msg_out_projection.weight"] = rnd(g.hidden_dim, g.num_heads * per_tower) * 0.2
        sd[pre + "msg_out_projection.bias"] = rnd(g.hidden_dim)
        for ln in ("mp_norm_layer", "boom_norm_layer"):
            sd[pre + ln + ".weight"] = rnd(g.hidden_dim) + 1.0
            sd[pre + ln + ".bias"] = rnd(g.hidden_dim)
        sd[pre + "boom_layer.linear1.weight"] = rnd(g.intermediate_dim, g.hidden_dim)
        sd[pre + "boom_layer.linear1.bias"] = rnd(g.intermediate_dim)
        sd[pre + "boom_layer.linear2.weight"] = rnd(g.hidden_dim, g.intermediate_dim) * 0.2
        sd[pre + "boom_layer.linear2.bias"] = rnd(g.hidden_dim)
    node_dim = (g.num_layers + 1) * g.hidden_dim if r.use_all_states else g.hidden_dim
    hid = r.num_heads * r.head_dim
    for pool in ("_weighted_mean_pooler", "_weighted_sum_pooler"):
        pre = f"readout.{pool}."
        sd[pre + "_scoring_module._layers.0.weight"] = rnd(hid, node_dim) * 0.3
        sd[pre + "_scoring_module._layers.0.bias"] = rnd(hid)
        sd[pre + "_scoring_module._layers.2.weight"] = rnd(r.num_heads, hid) * 0.3
        sd[pre + "_scoring_module._layers.2.bias"] = rnd(r.num_heads)
        sd[pre + "_transformation_mlp._layers.0.weight"] = rnd(hid, node_dim) * 0.3
        sd[pre + "_transformation_mlp._layers.0.bias"] = rnd(hid)
        sd[pre + "_transformation_mlp._layers.2.weight"] = rnd(hid, hid) * 0.3
        sd[pre + "_transformation_mlp._layers.2.bias"] = rnd(hid)
        sd[pre + "_combination_layer.weight"] = rnd(r.output_dim, hid)
    sd["readout._max_pooler._combination_layer.weight"] = rnd(r.output_dim, node_dim)
    sd["readout._combination_layer.weight"] = rnd(r.output_dim, 3 * r.output_dim)
    return {prefix + k: v for k, v in sd.items()}


def std_indicators(messages_per_block, adjacency_lists, num_nodes: int, m: int, bidirectional: bool = True):
    """The indicators [b_e^2 > mean^2] of the std aggregation from captured post-ReLU messages ([E_all, heads, 3 m] per block, e.g. a
    float32 device forward: ``TowerMessagePassing.capture``), evaluated in float64 - products of float32 numbers are exact there,
    which is also how the device kernels decide (csrc/pna.h)."""
    adj = [torch.cat((a, a.flip(1)), 0) if bidirectional else a for a in adjacency_lists]
    tg = torch.cat([a[:, 1] for a in adj]).cpu()
    cnt = torch.bincount(tg, minlength=num_nodes).clamp(min=1).double().view(num_nodes, 1, 1)
    out = []
    for msgs in messages_per_block:
        b = msgs.detach().double().cpu()[..., m:2 * m]
        mean = torch.zeros(num_nodes, *b.shape[1:], dtype=torch.float64).index_add_(0, tg, b) / cnt
        out.append(b.pow(2) > mean[tg].pow(2))
    return out
