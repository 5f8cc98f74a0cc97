"""GPU: run-to-run reproducibility of the hot path, bit for bit (SURVEY section 5 "determinism check by re-run bit-compare").

The reference's extractor sums with torch_scatter / ``index_add_`` (fs_mol/modules/gnn.py:203-244,
fs_mol/modules/graph_readout.py:238-252,289), floating-point atomics on a GPU; this build's per-node and per-graph sums run in
a fixed order (csrc/pna.h, csrc/readout.h) and the GP kernels reduce through fixed trees, so two runs on identical inputs
give identical bits - which is what lets the self-comparison tests of tests/test_gpu_f3.py assert ``torch.equal`` on FITTED
quantities (a converged float32 inner fit amplifies a 1e-7 difference of its inputs to 1e-4 on f_out).
Collected after the oracle / golden parity files (tests/conftest.py)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    return torch.device("cuda:0")


def _c3_tasks(n_tasks, ns, nq):
    from adkf_ift_amd.meta_batch import DKTBatch
    from test_gpu_gnn import _molecules

    g = torch.Generator().manual_seed(11)
    return [DKTBatch(_molecules(ns, 50 + 2 * t), torch.rand(ns, generator=g) > 0.5, torch.randn(ns, generator=g),
                     _molecules(nq, 51 + 2 * t), torch.rand(nq, generator=g) > 0.5, torch.randn(nq, generator=g))
            for t in range(n_tasks)]


def test_default_width_c3_meta_step_is_bit_reproducible(dev):
    """BASELINE config 3 (default 25 M-parameter GNN + ECFP + fc model): two ``model_meta_step`` runs from identical weights -
    losses, fitted phi, every parameter gradient and every updated parameter identical to the bit."""
    from adkf_ift_amd.meta_batch import collate_meta_batch, model_meta_step
    from adkf_ift_amd.models import ADKTModel, ADKTModelConfig
    from adkf_ift_amd.trainer import MetaStepConfig

    mb = collate_meta_batch(_c3_tasks(3, 16, 40)).to(dev)
    runs = []
    for _ in range(2):
        torch.manual_seed(3)
        model = ADKTModel(ADKTModelConfig()).to(dev)
        with torch.no_grad():
            for blk in model.graph_feature_extractor.gnn.gnn_blocks:
                blk.alpha.fill_(0.3)          # ReZero's 1e-7 at initialisation would hide the message passing
        params = list(model.feature_extractor_params())
        opt = torch.optim.SGD(params, lr=0.1)
        losses, phi = model_meta_step(model, opt, mb, MetaStepConfig(gp_kernel="matern", clip_value=1.0), check=True)
        torch.cuda.synchronize()
        runs.append((losses.clone(), phi.clone(), [p.grad.clone() for p in params if p.grad is not None], [p.detach().clone() for p in params]))
    (la, fa, ga, pa), (lb, fb, gb, pb) = runs
    assert torch.isfinite(la).all()
    assert torch.equal(la, lb) and torch.equal(fa, fb)
    assert max(float(g.abs().max()) for g in ga) > 0.0
    for x, y in zip(ga + pa, gb + pb):
        assert torch.equal(x, y)


def test_extractor_forward_backward_is_bit_reproducible_small_odd_shapes(dev):
    """The fused kernels on sizes that are no multiple of anything (3 heads x 5, 6-wide messages, isolated nodes, a single-atom
    graph, an empty edge type): forward and every parameter gradient identical over two runs."""
    from adkf_ift_amd.gnn import GraphFeatureExtractor
    from test_gnn import random_graphs, small_cfg

    batch = random_graphs(9, seed=13, empty_type=1).to(dev)
    batch.node_features = batch.node_features.float()
    outs = []
    for _ in range(2):
        torch.manual_seed(5)
        model = GraphFeatureExtractor(small_cfg()).to(dev)
        with torch.no_grad():
            for blk in model.gnn.gnn_blocks:
                blk.alpha.fill_(0.6)
        z = model(batch)
        (z * torch.linspace(-1.0, 1.0, z.numel(), device=dev).view_as(z)).sum().backward()
        torch.cuda.synchronize()
        outs.append([z.detach().clone()] + [p.grad.clone() for n, p in model.named_parameters() if p.grad is not None])
    assert len(outs[0]) > 10
    for x, y in zip(*outs):
        assert torch.equal(x, y)


def test_gp_section_is_bit_reproducible_at_c2_shape(dev):
    """fit -> IFT hypergradient of 8 tasks at the C2 shape (N = N_q = 128, d = 256), twice in fresh workspaces: phi*, f_in, f_out
    and both cotangents identical to the bit (fs_mol/utils/adaptive_dkt_utils.py:91, fs_mol/utils/cauchy_hypergradient.py:120-161)."""
    from adkf_ift_amd import gp_ops
    from adkf_ift_amd.synthetic import make_tasks

    tasks = make_tasks(8, 128, 256)
    Zs, Zq = tasks.features()
    outs = []
    for _ in range(2):
        phi, pri, _ = gp_ops.init_params(Zs.to(dev))
        b = gp_ops.GPBatch(Zs.to(dev), tasks.y_s.to(dev), pri, "rbf", Z_q=Zq.to(dev), y_q=tasks.y_q.to(dev))
        phi, f, gn, ne, info = gp_ops.fit(b, phi, max_evals=200)
        gp_ops.check_info(info)
        out = gp_ops.ift_hypergrad(b, phi)
        gp_ops.check_info(out["info"])
        torch.cuda.synchronize()
        outs.append([phi.clone(), f.clone(), out["f_out"].clone(), out["dZ_s"].clone(), out["dZ_q"].clone()])
    for x, y in zip(*outs):
        assert torch.equal(x, y)
