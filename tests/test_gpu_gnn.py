"""GPU: row a1 (GNN feature extractor) at the REFERENCE'S WIDTH against oracle/gnn_oracle.py, and BASELINE config C3
(full default deep-kernel model, 16-shot molecular tasks) against the per-task float64 oracle loop.

fs_mol/modules/gnn.py:401-515,530-556 (hidden 128, 4 towers x 64, 10 PNA layers, BOOM 1024) and
fs_mol/modules/graph_readout.py:119-177 (12 heads x 64 over all 11 node states); head fs_mol/models/adaptive_dkt.py:50-65.
The device side runs float32 with the fused HIP message / aggregation kernels (csrc/pna.h); the oracle side is the naive
module-by-module restatement in float64 on the CPU, parameters under the reference's names."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    return torch.device("cuda:0")


def test_default_width_extractor_forward_and_gradients_vs_oracle(dev):
    from adkf_ift_amd.gnn import GraphFeatureExtractor, GraphFeatureExtractorConfig
    from oracle import gnn_oracle as GO
    from test_gnn import grads_under_reference_names, random_graphs, unit_gain_reference_state_dict

    cfg = GraphFeatureExtractorConfig()
    sd = {k: v.requires_grad_(True) for k, v in unit_gain_reference_state_dict(cfg, seed=2).items()}
    batch = random_graphs(40, seed=11)          # isolated nodes and a single-atom graph included
    want = GO.graph_feature_extractor(batch, sd, cfg)
    model = GraphFeatureExtractor(cfg)
    model.load_reference_state_dict({k: v.detach().float() for k, v in sd.items()})
    model = model.to(dev)
    b32 = batch.to(dev)
    b32.node_features = b32.node_features.float()
    got = model(b32)
    assert got.shape == (40, 512)
    err = (got.double().cpu() - want).abs().max().item() / want.abs().max().item()
    assert err <= 2e-5, err
    w = torch.randn(want.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(1))
    (want * w).sum().backward()
    (got * w.float().to(dev)).sum().backward()
    mine = grads_under_reference_names(model)
    scale = max(v.grad.abs().max().item() for v in sd.values() if v.grad is not None)
    # The reference's std aggregation sqrt(sum_e relu(b_e^2 - mean^2) + 1e-7) (fs_mol/modules/gnn.py:231-240) has slope
    # 1 / (2 sqrt(1e-7)) = 1581 at zero variance, so for nearly equal incoming messages the float32 rounding of the MESSAGES
    # (1e-7 b^2 against a floor of 1e-7) moves the gradient in any float32 implementation.  The fused kernels (csrc/pna.h) form
    # mean, deviations and the indicators in float64 - exact for float32 inputs - which leaves only that input rounding: 2.9e-4
    # of the largest gradient entry here, against 5.7e-4 with float32 accumulation (round 2) and 8.5e-4 for float32 PyTorch on
    # the CPU (the yardstick below, printed, no longer part of the tolerance).  Fixed bound: 4e-4.
    cpu32 = GraphFeatureExtractor(cfg)
    cpu32.load_reference_state_dict({k: v.detach().float() for k, v in sd.items()})
    c32 = batch.to("cpu")
    c32.node_features = c32.node_features.float()
    (cpu32(c32) * w.float()).sum().backward()
    yard = grads_under_reference_names(cpu32)
    e32 = max((yard[k].double() - v.grad).abs().max().item() / scale for k, v in sd.items() if v.grad is not None)
    tol = 4e-4
    worst = 0.0
    for k, v in sd.items():
        if v.grad is None:
            continue
        e = (mine[k].double().cpu() - v.grad).abs().max().item() / scale
        worst = max(worst, e)
        assert e <= tol, (k, e, e32)
    print("default-width extractor: forward rel err %.2e; worst parameter-gradient err %.2e of the largest entry "
          "(float32 PyTorch on the CPU: %.2e)" % (err, worst, e32))


@pytest.mark.parametrize("kind,empty_type", [("PNA", None), ("PNA", 1), ("MultiAggr", 2)])
def test_fused_kernels_on_odd_shapes_vs_cpu_float64(dev, kind, empty_type):
    """The order-fixed kernels of round 4 (message-function backward: d cat + CSR gather, chunk partials for d W / d b,
    csrc/pna.h; read-out pooling forward / backward, csrc/readout.h) on sizes that are a multiple of nothing - 4 towers x 6-wide
    messages, 3 read-out heads x 5, isolated nodes, a single-atom graph, an edge type without edges - against the SAME module
    evaluated in float64 on the CPU through PyTorch's own index_add_ / scatter_reduce_ / autograd
    (fs_mol/modules/gnn.py:203-244, fs_mol/modules/graph_readout.py:236-252,289)."""
    from adkf_ift_amd.gnn import GraphFeatureExtractor
    from test_gnn import random_graphs, small_cfg

    cfg = small_cfg(kind)
    batch = random_graphs(11, seed=17, empty_type=empty_type)
    torch.manual_seed(9)
    ref = GraphFeatureExtractor(cfg).double()
    with torch.no_grad():
        for blk in ref.gnn.gnn_blocks:
            blk.alpha.fill_(0.6)
    got = GraphFeatureExtractor(cfg)
    got.load_state_dict({k: v.float() for k, v in ref.state_dict().items()})
    got = got.to(dev)
    b32 = batch.to(dev)
    b32.node_features = b32.node_features.float()
    want = ref(batch)
    z = got(b32)
    w = torch.randn(want.shape, dtype=torch.float64, generator=torch.Generator().manual_seed(2))
    (want * w).sum().backward()
    (z * w.float().to(dev)).sum().backward()
    assert (z.double().cpu() - want).abs().max().item() <= 2e-5 * want.abs().max().item()
    named = dict(got.named_parameters())
    scale = max(p.grad.abs().max().item() for p in ref.parameters() if p.grad is not None)
    for n, p in ref.named_parameters():
        if p.grad is None:
            assert named[n].grad is None, n
            continue
        e = (named[n].grad.double().cpu() - p.grad).abs().max().item() / scale
        assert e <= 5e-5, (n, e)


def _molecules(n, seed):
    from adkf_ift_amd.meta_batch import MoleculeFeatures
    from test_gnn import random_graphs
    g = torch.Generator().manual_seed(1000 + seed)
    gb = random_graphs(n, seed=seed)
    return MoleculeFeatures(gb.node_features.float(), gb.adjacency_lists, gb.node_to_graph, gb.num_graphs,
                            torch.poisson(torch.full((n, 2048), 0.03), generator=g), torch.randn(n, 42, generator=g))


def test_c3_default_model_meta_step_vs_per_task_oracle_loop(dev):
    """BASELINE config 3: GNN + ECFP -> fc(2560 -> 2048 -> 2048) -> GP fit + IFT hypergradient, 2 tasks of 16 support +
    32 query molecules, through ``model_meta_step`` (one extractor forward/backward for both tasks).
    Expected: per task, oracle features (gnn_oracle + fc, float64) -> d f_out/dZ - v^T d^2 f_in/dphi dZ from the float64
    autograd GP oracle AT THE DEVICE'S FITTED phi -> one backward through the oracle extractor; mean over tasks
    (fs_mol/utils/adaptive_dkt_utils.py:361-407).  At the feature level that total derivative is exactly what the
    reference's cauchy_hypergradient returns (pinned by the linmap fixtures); chaining it through the extractor is the
    chain rule."""
    from adkf_ift_amd.meta_batch import DKTBatch, collate_meta_batch, model_meta_step
    from adkf_ift_amd.models import ADKTModel, ADKTModelConfig
    from adkf_ift_amd.trainer import MetaStepConfig
    from oracle import gnn_oracle as GO
    from oracle import gp_oracle as O
    from test_gnn import grads_under_reference_names, unit_gain_reference_state_dict

    torch.manual_seed(0)
    mcfg = ADKTModelConfig()                      # reference defaults
    model = ADKTModel(mcfg)
    gcfg = model.graph_feature_extractor.config
    sd = unit_gain_reference_state_dict(gcfg, seed=5)
    model.graph_feature_extractor.load_reference_state_dict({k: v.float() for k, v in sd.items()})
    model = model.to(dev)
    g = torch.Generator().manual_seed(3)
    tasks = []
    for t in range(2):
        ns, nq = 16, 32
        tasks.append(DKTBatch(_molecules(ns, 20 + 2 * t), torch.rand(ns, generator=g) > 0.5, torch.randn(ns, generator=g),
                              _molecules(nq, 21 + 2 * t), torch.rand(nq, generator=g) > 0.5, torch.randn(nq, generator=g)))
    mb = collate_meta_batch(tasks).to(dev)
    cfg = MetaStepConfig(gp_kernel="matern", clip_value=None)
    losses, phi = model_meta_step(model, None, mb, cfg, check=True)

    # ---- oracle side (CPU, float64) ----
    sd64 = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    fc = [p.detach().double().cpu().requires_grad_(True) for p in model.fc.parameters()]     # W1, b1, W2, b2

    def features(part):
        gb = part.graph()
        gb.node_features = gb.node_features.double()
        h = GO.graph_feature_extractor(gb, sd64, gcfg)
        x = torch.cat([h, part.fingerprints.double()], dim=1)
        return torch.relu(x @ fc[0].T + fc[1]) @ fc[2].T + fc[3]

    T = len(tasks)
    want_losses = []
    for t, b in enumerate(tasks):
        Zs, Zq = features(b.support_features), features(b.query_features)
        ys, yq = (b.support_labels.double() - 0.5) * 2, (b.query_labels.double() - 0.5) * 2
        _, pri = O.init_phi(Zs.detach(), False, True)
        q = O.full_reference_quantities(Zs.detach(), ys, Zq.detach(), yq, phi[t].double().cpu(), pri, O.KERNEL_MATERN52)
        want_losses.append(q["f_out"] / b.num_query_samples)
        torch.autograd.backward([Zs, Zq], [torch.tensor(q["dZs_total"]) / T, torch.tensor(q["dZq_total"]) / T])
    assert np.abs(losses.cpu().numpy() - np.array(want_losses)).max() <= 1e-4 * np.abs(want_losses).max()
    mine = grads_under_reference_names(model.graph_feature_extractor)
    # Bound: 9e-4 of the largest gradient entry, set from the REPRODUCIBLE value of this test: 7.38e-4 (round 4: the extractor has
    # no floating-point atomics any more, tests/test_gpu_determinism.py, so the number no longer moves between runs - round 3 saw
    # 2.9e-4 and 7.4e-4 on the same tree and asserted 1e-3).  The extractor alone is at 2.9e-4 (test above).  What is left is the
    # std aggregation of the reference, whose gradient is DISCONTINUOUS in its inputs (the indicator [b_e^2 > mean^2] of
    # fs_mol/modules/gnn.py:231-240, times a slope of up to 1581): the (layer, node, tower, feature) entries where the float32
    # forward and the float64 oracle fall on different sides are counted and printed below.
    C3_TOL = 9e-4
    scale = max(max(v.grad.abs().max().item() for v in sd64.values() if v.grad is not None), max(p.grad.abs().max().item() for p in fc))
    worst = 0.0
    for k, v in sd64.items():
        if v.grad is None:
            continue
        e = (mine[k].double().cpu() - v.grad).abs().max().item() / scale
        worst = max(worst, e)
        assert e <= C3_TOL, (k, e)
    for p, r in zip(model.fc.parameters(), fc):
        e = (p.grad.double().cpu() - r.grad).abs().max().item() / scale
        worst = max(worst, e)
        assert e <= C3_TOL, e
    print("C3 default model: worst theta.grad error %.2e of the largest entry" % worst)
    # ---- which indicators of the std aggregation differ between the float32 device forward and a float64 forward ----
    from adkf_ift_amd.gnn import GraphFeatureExtractor
    twin = GraphFeatureExtractor(gcfg).double()
    twin.load_reference_state_dict(sd)
    caps64, caps32 = [], []
    for blk64, blk32 in zip(twin.gnn.gnn_blocks, model.graph_feature_extractor.gnn.gnn_blocks):
        blk64.mp.capture, blk32.mp.capture = caps64, caps32
    mols = mb.molecules
    with torch.no_grad():
        g64 = mols.graph().to("cpu")
        g64.node_features = g64.node_features.double()
        twin(g64)
        model.graph_feature_extractor(mols.graph())
    for blk in model.graph_feature_extractor.gnn.gnn_blocks:
        blk.mp.capture = None
    adj = [torch.cat((a, a.flip(1)), 0) for a in g64.adjacency_lists]
    tg = torch.cat([a[:, 1] for a in adj])
    V, m = g64.node_features.shape[0], gcfg.gnn_config.per_head_dim
    cnt = torch.bincount(tg, minlength=V).clamp(min=1).double().view(V, 1, 1)

    def indicator(msgs):
        b = msgs.double().cpu()[..., m:2 * m]
        mean = torch.zeros(V, *b.shape[1:], dtype=torch.float64).index_add_(0, tg, b) / cnt
        return b.pow(2) > mean[tg].pow(2)
    flips = [int((indicator(a) != indicator(b)).sum()) for a, b in zip(caps64, caps32)]
    total = caps64[0][..., m:2 * m].numel()
    print("C3 default model: std-aggregation indicators that differ between the float32 device forward and a float64 forward, "
          "per layer (of %d each): %s" % (total, flips))
