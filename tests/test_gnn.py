"""CPU: the re-authored GNN feature extractor (fused towers, native scatter ops, one GEMM for the read-out MLP heads)
equals the naive module-by-module restatement of the reference (oracle/gnn_oracle.py) evaluated from a state dict
with the REFERENCE's parameter names - which also tests ``load_reference_state_dict``.  float64, random small
graphs with isolated nodes, an empty edge type and a single-node graph."""
import numpy as np
import pytest
import torch

from adkf_ift_amd.gnn import (GNNConfig, GraphBatch, GraphFeatureExtractor, GraphFeatureExtractorConfig,
                              GraphReadoutConfig, concat_graph_batches)
from oracle import gnn_oracle as GO


def random_graphs(num_graphs, seed, num_edge_types=3, feat=32, empty_type=None):
    g = torch.Generator().manual_seed(seed)
    sizes = [1] + [int(torch.randint(2, 9, (1,), generator=g)) for _ in range(num_graphs - 1)]  # first graph: a single atom
    feats, n2g, adj = [], [], [[] for _ in range(num_edge_types)]
    v0 = 0
    for gi, n in enumerate(sizes):
        feats.append(torch.randn(n, feat, generator=g, dtype=torch.float64))
        n2g += [gi] * n
        for t in range(num_edge_types):
            if t == empty_type or n < 2:
                continue
            e = int(torch.randint(0, n + 1, (1,), generator=g))
            if e:
                src = torch.randint(0, n - 1, (e,), generator=g)     # node n-1 never a source ...
                tgt = torch.randint(0, n - 1, (e,), generator=g)     # ... nor a target: isolated node
                adj[t].append(torch.stack([src, tgt], 1) + v0)
        v0 += n
    adj = [torch.cat(a) if a else torch.zeros(0, 2, dtype=torch.long) for a in adj]
    return GraphBatch(torch.cat(feats), adj, torch.tensor(n2g), num_graphs)


def small_cfg(kind="PNA", depth=1, all_states=True):
    return GraphFeatureExtractorConfig(
        gnn_config=GNNConfig(type=kind, hidden_dim=16, num_heads=4, per_head_dim=6, intermediate_dim=24,
                             num_layers=3, message_function_depth=depth),
        readout_config=GraphReadoutConfig(use_all_states=all_states, num_heads=3, head_dim=5, output_dim=10))


@pytest.mark.parametrize("kind,depth,all_states,empty_type", [("PNA", 1, True, None), ("PNA", 2, True, 1),
                                                             ("MultiAggr", 1, False, None), ("Plain", 1, True, 2)])
def test_extractor_matches_reference_restatement(kind, depth, all_states, empty_type):
    cfg = small_cfg(kind, depth, all_states)
    batch = random_graphs(6, seed=3, empty_type=empty_type)
    ref_sd = GO.random_reference_state_dict(cfg, seed=1)
    model = GraphFeatureExtractor(cfg).double()
    model.load_reference_state_dict(ref_sd)
    got = model(batch)
    want = GO.graph_feature_extractor(batch, ref_sd, cfg)
    assert got.shape == (6, 10)
    assert torch.allclose(got, want, rtol=1e-10, atol=1e-10), (got - want).abs().max()
    # gradients flow to every parameter that the reference forward uses (mp_norm_layer is unused there too)
    got.sum().backward()
    for n, p in model.named_parameters():
        if "mp_norm_layer" in n:
            assert p.grad is None
        else:
            assert p.grad is not None and torch.isfinite(p.grad).all(), n


def test_concatenated_tasks_equal_separate_forwards():
    """One disconnected graph for many tasks == the per-task forwards (what lets a meta-batch cost ONE forward)."""
    cfg = small_cfg()
    model = GraphFeatureExtractor(cfg).double()
    with torch.no_grad():
        for blk in model.gnn.gnn_blocks:
            blk.alpha.fill_(0.7)
    parts = [random_graphs(4, seed=s) for s in (5, 6, 7)]
    whole = model(concat_graph_batches(parts))
    sep = torch.cat([model(p) for p in parts])
    assert torch.allclose(whole, sep, rtol=1e-12, atol=1e-12)


def test_default_config_is_the_reference_cli_default():
    cfg = GraphFeatureExtractorConfig()
    model = GraphFeatureExtractor(cfg)
    n_gnn = sum(p.numel() for n, p in model.named_parameters() if n.startswith("gnn."))
    n_ro = sum(p.numel() for n, p in model.named_parameters() if n.startswith("readout."))
    # GNN ~ 8.1 M, read-out ~ 7.8 M parameters
    assert 7.5e6 < n_gnn < 8.7e6 and 7.2e6 < n_ro < 8.4e6, (n_gnn, n_ro)
    batch = random_graphs(5, seed=0)
    out = model(GraphBatch(batch.node_features.float(), batch.adjacency_lists, batch.node_to_graph, batch.num_graphs))
    assert out.shape == (5, 512) and torch.isfinite(out).all()


def unit_gain_reference_state_dict(cfg, seed):
    """GO.random_reference_state_dict rescaled to unit-gain layers (weights ~ 1/sqrt(fan_in), alpha ~ 0.3): keeps the
    activations O(1) through ten layers at the reference's default width (hidden 128, 4 towers x 64, 3072-wide
    aggregation), where N(0, 0.3^2) weights would overflow float32."""
    sd = GO.random_reference_state_dict(cfg, seed=seed)
    for k, v in sd.items():
        if k.endswith(".weight") and v.dim() == 2 and "norm" not in k:
            sd[k] = v / (0.3 * v.shape[1] ** 0.5)
        if k.endswith("alpha"):
            sd[k] = v * 0.4
    return sd


def grads_under_reference_names(model):
    """d out / d parameters of ``model`` re-keyed by the REFERENCE's parameter names: the name mapping is a linear
    re-stacking, so it applies to gradients unchanged (a scratch copy whose parameters hold the gradients is exported)."""
    import copy
    g = copy.deepcopy(model)
    with torch.no_grad():
        for q, p in zip(g.parameters(), model.parameters()):
            q.copy_(p.grad if p.grad is not None else torch.zeros_like(p))
    return g.reference_state_dict()


def test_default_width_matches_reference_restatement_with_gradients():
    """fs_mol/modules/gnn.py:401-515 at the CLI defaults (hidden 128, 4 towers x 64, 10 layers, BOOM 1024) and
    fs_mol/modules/graph_readout.py:119-177 (12 heads x 64, all 11 states = 1408-wide nodes): forward AND parameter
    gradients of the re-authored extractor against the naive restatement, float64."""
    cfg = GraphFeatureExtractorConfig()
    sd = {k: v.requires_grad_(True) for k, v in unit_gain_reference_state_dict(cfg, seed=2).items()}
    batch = random_graphs(40, seed=11)
    model = GraphFeatureExtractor(cfg).double()
    model.load_reference_state_dict({k: v.detach() for k, v in sd.items()})
    got = model(batch)
    want = GO.graph_feature_extractor(batch, sd, cfg)
    assert got.shape == (40, 512)
    assert (got - want).abs().max().item() <= 1e-10 * want.abs().max().item()
    w = torch.randn(want.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    (got * w).sum().backward()
    (want * w).sum().backward()
    mine = grads_under_reference_names(model)
    scale = max(v.grad.abs().max().item() for v in sd.values() if v.grad is not None)
    for k, v in sd.items():
        if v.grad is None:
            assert "mp_norm_layer" in k, k      # constructed but not applied in the reference forward (gnn.py:477-515)
            continue
        assert (mine[k] - v.grad).abs().max().item() <= 1e-9 * scale, k


def test_batchnorm_running_statistics_round_trip():
    cfg = small_cfg()
    cfg.output_norm = "batch"
    model = GraphFeatureExtractor(cfg).double()
    model.train()
    for s in (1, 2, 3):
        model(random_graphs(6, seed=s))          # moves running_mean / running_var away from 0 / 1
    ref = model.reference_state_dict()
    assert "graph_feature_extractor.final_norm_layer.running_mean" in ref
    other = GraphFeatureExtractor(cfg).double()
    other.load_reference_state_dict(ref)
    model.eval(), other.eval()
    b = random_graphs(5, seed=9)
    assert torch.equal(model(b), other(b))
    del ref["graph_feature_extractor.final_norm_layer.running_var"]
    with pytest.raises(KeyError):
        GraphFeatureExtractor(cfg).double().load_reference_state_dict(ref)


def test_readout_projection_of_pooled_hidden_states_is_the_same_algebra():
    """What the device path does by default (csrc/readout.h, k_readout_h_*): pool the HIDDEN activations of the value MLPs per
    head, p[h, g, :] = sum_v w[v, h] r_v, then project, W2[h] p + b2[h] sum_v w[v, h] - against the module's own order (project every
    node, then pool: fs_mol/modules/graph_readout.py:219-223, 242-252).  Pure torch in float64 on the CPU: the identity behind
    ``CombinedGraphReadout._project_pooled``, for the softmax-weighted and the sigmoid-weighted head, an empty graph included."""
    from adkf_ift_amd.gnn import CombinedGraphReadout, _segment_softmax

    torch.manual_seed(4)
    nh, hd, D, sizes = 3, 5, 11, [4, 0, 7, 1]
    n2g = torch.cat([torch.full((n,), g, dtype=torch.long) for g, n in enumerate(sizes)])
    G, V, hid = len(sizes), int(n2g.shape[0]), nh * hd
    ro = CombinedGraphReadout(D, 9, nh, hd).double()
    x = torch.randn(V, D, dtype=torch.float64)
    h = torch.relu(ro.first(x))
    h_ms, h_mv, h_ss, h_sv = h.split(hid, dim=1)
    for scores, hval, layer, softmax in ((ro.mean_score_out(h_ms), h_mv, ro.mean_value_out, True), (ro.sum_score_out(h_ss), h_sv, ro.sum_value_out, False)):
        w = _segment_softmax(scores, n2g, G) if softmax else torch.sigmoid(scores)                      # [V, nh]
        want = torch.zeros(G, hid, dtype=torch.float64).index_add(0, n2g, (w.unsqueeze(-1) * layer(hval).view(V, nh, hd)).reshape(V, hid))
        p = torch.zeros(nh, G, hid, dtype=torch.float64)
        for hh in range(nh):
            p[hh].index_add_(0, n2g, w[:, hh:hh + 1] * hval)
        wtot = torch.zeros(G, nh, dtype=torch.float64).index_add(0, n2g, w)
        got = ro._project_pooled(p, wtot, layer)
        assert torch.allclose(got, want, rtol=1e-12, atol=1e-13)
        if softmax:   # the weights of a graph sum to one (zero for the empty graph): the kernel returns exactly that
            assert torch.allclose(wtot, torch.tensor([[1.0] * nh if n > 0 else [0.0] * nh for n in sizes], dtype=torch.float64), atol=1e-12)


def c3_test_graphs():
    """The molecules of tests/test_gpu_gnn.py::test_c3_default_model_meta_step_vs_per_task_oracle_loop as ONE disconnected graph
    (2 tasks x (16 support + 32 query), graph seeds 20 ... 23)."""
    parts = [random_graphs(n, seed=s) for n, s in ((16, 20), (32, 21), (16, 22), (32, 23))]
    feats, adj, n2g, v0, g0 = [], [[], [], []], [], 0, 0
    for gb in parts:
        feats.append(gb.node_features.float().double())   # (the GPU test's molecules carry float32 node features)
        for t in range(3):
            adj[t].append(gb.adjacency_lists[t] + v0)
        n2g.append(gb.node_to_graph + g0)
        v0 += gb.node_features.shape[0]
        g0 += gb.num_graphs
    return GraphBatch(torch.cat(feats), [torch.cat(a) for a in adj], torch.cat(n2g), g0)


@pytest.mark.parametrize("which,floor_lo,floor_hi", [("extractor", 1.5e-4, 4.5e-4), ("c3", 3e-4, 8e-4)])
def test_float32_node_states_set_the_gradient_error_floor(which, floor_lo, floor_hi):
    """WHERE the 3e-4 ... 1e-3 (of the largest entry) between ANY float32 evaluation of the default-width extractor's parameter
    gradients and the float64 restatement comes from, on the exact inputs of the two default-width GPU tests (tests/test_gpu_gnn.py).
    Everything below runs in float64; ONE kind of intermediate at a time is rounded to float32 (straight-through, so the gradient
    still flows):

      * the post-ReLU messages of every tower and block (what a float32 message GEMM rounds):          2e-5
      * the aggregates sum | mean | std | max of every block:                                          5e-6
      * the NODE STATES between blocks (what any implementation that stores them in float32 rounds):   2.8e-4 / 5.3e-4
      * float32 PyTorch on the CPU, everything in float32 (the reference's arithmetic):                1.0e-3 / 1.1e-3

    So the error is not a property of a kernel: 6e-8 relative on the stored node states is amplified 5 000 x on these two draws
    (the reference's std aggregation sqrt(sum relu(b^2 - mean^2) + 1e-7), fs_mol/modules/gnn.py:231-240, at a node whose incoming
    messages are nearly equal; other seeds give floors of 5e-6 ... 7e-5), and no float32-state implementation can be closer to
    float64 than that floor.  Round 4 blamed the rounding of the MESSAGES (it is 10 x too small) and before that the indicator flips
    (< 1 %).  The GPU tests hold the device to FIXED bounds of about twice this floor: 6e-4 and 1.5e-3."""
    cfg = GraphFeatureExtractorConfig()
    if which == "extractor":
        batch, sd0, wseed = random_graphs(40, seed=11), unit_gain_reference_state_dict(cfg, seed=2), 1
    else:
        batch, sd0, wseed = c3_test_graphs(), unit_gain_reference_state_dict(cfg, seed=5), 1
    w = torch.randn(batch.num_graphs, 512, dtype=torch.float64, generator=torch.Generator().manual_seed(wseed))
    through = lambda o: o + (o.float().double() - o).detach()

    def run(dtype, hooks=None):
        net = GraphFeatureExtractor(cfg)
        net.load_reference_state_dict({k: v.detach().float() for k, v in sd0.items()})
        net = net.to(dtype)
        b = batch.to("cpu")
        b.node_features = b.node_features.to(dtype)
        hs = hooks(net) if hooks else []
        (net(b) * w.to(dtype)).sum().backward()
        for h in hs:
            h.remove()
        return {k: v.double() for k, v in grads_under_reference_names(net).items()}

    g64 = run(torch.float64)
    scale = max(g.abs().max().item() for g in g64.values())
    err = lambda g: max((g[k] - g64[k]).abs().max().item() / scale for k in g64)
    e32 = err(run(torch.float32))
    e_states = err(run(torch.float64, lambda net: [blk.register_forward_hook(lambda m, i, o: through(o)) for blk in net.gnn.gnn_blocks]))
    e_aggr = err(run(torch.float64, lambda net: [blk.mp.register_forward_hook(lambda m, i, o: through(o)) for blk in net.gnn.gnn_blocks]))
    print("%s inputs: float32 everywhere %.2e | float64 with float32 node states %.2e | ... with float32 aggregates %.2e" % (which, e32, e_states, e_aggr))
    assert floor_lo <= e_states <= floor_hi, e_states
    assert e_aggr <= 0.1 * e_states, (e_aggr, e_states)
    assert e32 >= e_states, (e32, e_states)
    if which == "extractor":   # the messages, through the naive restatement's hook (the module computes them inside one function)
        sd = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
        (GO.graph_feature_extractor(batch, sd, cfg) * w).sum().backward()
        ref = {k: v.grad for k, v in sd.items() if v.grad is not None}
        sd2 = {k: v.clone().requires_grad_(True) for k, v in sd0.items()}
        (GO.graph_feature_extractor(batch, sd2, cfg, msg_hook=lambda name, m, t: through(m)) * w).sum().backward()
        sc = max(g.abs().max().item() for g in ref.values())
        e_msg = max((sd2[k].grad - g).abs().max().item() / sc for k, g in ref.items())
        print("          ... with float32 messages %.2e" % e_msg)
        assert e_msg <= 0.2 * e_states, (e_msg, e_states)
