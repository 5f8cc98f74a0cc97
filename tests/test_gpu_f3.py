"""GPU, SURVEY section 8 row f3 on the device path.

(1) numpy ``DKTBatch`` / ``FSMolBatch``-shaped objects (the field layout of fs_mol/data/dkt.py:25-46 and
    fs_mol/data/fsmol_batcher.py:22-94, restated in tests/test_meta_batch.py) -> ``dkt_batch_from_fsmol`` ->
    ``collate_meta_batch`` -> ``model_meta_step`` give, BIT FOR BIT, what the torch-built batch of the same molecules gives.
(2) a checkpoint with the reference's names (``save_model``, fs_mol/utils/adaptive_dkt_utils.py:221-259; loaded as
    :305-327 does) written by checkpoint.py, reloaded through ``load_reference_checkpoint`` into an ``ADKTModel``:
    ``evaluate.meta_test`` predictions are unchanged, bit for bit.
Every number comes from libadkf_gp.so + the fused extractor kernels; nothing here reads /root/reference."""
from dataclasses import dataclass
from typing import List

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@dataclass(frozen=True)
class RefMolecules:                      # FSMolBatch + the two MoleculeDKTFeatures fields (field layout only)
    num_graphs: int
    num_nodes: int
    num_edges: int
    node_features: np.ndarray
    adjacency_lists: List[np.ndarray]
    edge_features: List[np.ndarray]
    node_to_graph: np.ndarray
    fingerprints: np.ndarray
    descriptors: np.ndarray


@dataclass(frozen=True)
class RefDKTBatch:
    support_features: RefMolecules
    support_labels: np.ndarray
    support_numeric_labels: np.ndarray
    query_features: RefMolecules
    query_labels: np.ndarray
    query_numeric_labels: np.ndarray


def _ref_and_torch_task(ns, nq, seed):
    """The same few-shot task twice: as the reference's numpy objects and as this package's torch DKTBatch."""
    from adkf_ift_amd.meta_batch import DKTBatch, MoleculeFeatures
    from test_gnn import random_graphs

    rng = np.random.default_rng(seed)

    def part(n, s):
        gb = random_graphs(n, seed=s)
        fp = rng.poisson(0.05, (n, 2048)).astype(np.int32)
        desc = rng.normal(size=(n, 42)).astype(np.float32)
        adj = [a.numpy().astype(np.int32) for a in gb.adjacency_lists]
        ref = RefMolecules(n, gb.node_features.shape[0], sum(a.shape[0] for a in adj), gb.node_features.numpy().astype(np.float32), adj,
                           [np.zeros((a.shape[0], 0), np.float32) for a in adj], gb.node_to_graph.numpy().astype(np.int64), fp, desc)
        tor = MoleculeFeatures(gb.node_features.float(), gb.adjacency_lists, gb.node_to_graph, gb.num_graphs,
                               torch.as_tensor(fp).float(), torch.as_tensor(desc))
        return ref, tor

    rs, ts = part(ns, 2 * seed)
    rq, tq = part(nq, 2 * seed + 1)
    ls, lq = rng.random(ns) > 0.5, rng.random(nq) > 0.5
    ls[0], ls[1] = True, False                                   # both classes in every support set
    vs, vq = rng.normal(size=ns).astype(np.float32), rng.normal(size=nq).astype(np.float32)
    ref = RefDKTBatch(rs, ls, vs, rq, lq, vq)
    tor = DKTBatch(ts, torch.as_tensor(ls), torch.as_tensor(vs), tq, torch.as_tensor(lq), torch.as_tensor(vq))
    return ref, tor


def _model(dev, numeric=False):
    from adkf_ift_amd.models import ADKTModel
    from test_meta_batch import small_model

    torch.manual_seed(4)
    model = ADKTModel(small_model(numeric)).to(dev)
    with torch.no_grad():
        for blk in model.graph_feature_extractor.gnn.gnn_blocks:
            blk.alpha.fill_(0.5)
    return model


def test_reference_numpy_batches_through_the_device_meta_step():
    from adkf_ift_amd.meta_batch import collate_meta_batch, dkt_batch_from_fsmol, model_meta_step

    dev = torch.device("cuda:0")
    pairs = [_ref_and_torch_task(16, 40, 31), _ref_and_torch_task(13, 9, 32), _ref_and_torch_task(16, 130, 33)]
    results = []
    for which in (0, 1):                                            # 0: through the adapter, 1: torch-built
        model = _model(dev)
        opt = torch.optim.SGD(list(model.feature_extractor_params()), lr=0.1)
        tasks = [dkt_batch_from_fsmol(p[0]) if which == 0 else p[1] for p in pairs]
        mb = collate_meta_batch(tasks).to(dev)
        losses, _ = model_meta_step(model, opt, mb, check=True)
        torch.cuda.synchronize()
        results.append((losses.cpu(), [p.detach().cpu().clone() for p in model.feature_extractor_params()],
                        [p.grad.detach().cpu().clone() for p in model.feature_extractor_params() if p.grad is not None]))
    (la, pa, ga), (lb, pb, gb) = results
    assert torch.isfinite(la).all() and la.shape == (3,)
    # The adapter only changes the container, so both runs are the same arithmetic on the same numbers - and, since round 4, to
    # the bit: the extractor's per-graph and per-node sums run in a fixed order (csrc/readout.h, csrc/pna.h: no floating-point
    # atomics), so the inner fit sees identical features and stops at the identical point.
    assert torch.equal(la, lb), (la, lb)
    assert len(ga) == len(gb) > 0
    assert max(float(g.abs().max()) for g in gb) > 0.0              # a step was taken
    for x, y in zip(ga, gb):
        assert torch.equal(x, y)
    for x, y in zip(pa, pb):
        assert torch.equal(x, y)


def test_reference_named_checkpoint_through_meta_test(tmp_path):
    from adkf_ift_amd import checkpoint as CK
    from adkf_ift_amd import evaluate as E
    from adkf_ift_amd.meta_batch import collate_meta_batch, dkt_batch_from_fsmol

    dev = torch.device("cuda:0")
    model = _model(dev)
    tasks = [dkt_batch_from_fsmol(_ref_and_torch_task(16, 40, 41)[0]), dkt_batch_from_fsmol(_ref_and_torch_task(16, 70, 42)[0])]
    mb = collate_meta_batch(tasks).to(dev)
    preds0, var0, phi0, _ = E.meta_test(model, mb, want_var=True)
    path = str(tmp_path / "best_validation.pt")
    CK.save_model(model, path, optimizer=torch.optim.Adam(model.feature_extractor_params(), 1e-4), epoch=3)
    sd = torch.load(path, map_location="cpu", weights_only=False)["model_state_dict"]
    # the reference's parameter names: per-tower extractor layers, the fc head, the GP copies
    assert any(k.startswith("graph_feature_extractor.gnn.gnn_blocks.0.") for k in sd) and "fc.0.weight" in sd
    assert "gp_model.covar_module.raw_outputscale" in sd
    model2, ck = CK.load_reference_checkpoint(path)
    assert ck["epoch"] == 3
    model2 = model2.to(dev)
    preds1, var1, phi1, _ = E.meta_test(model2, mb, want_var=True)
    torch.cuda.synchronize()
    # the same weights under the reference's names and back: the same extractor to the bit, hence the same fitted GP and the
    # same predictions to the bit (the path has no floating-point atomics since round 4)
    from adkf_ift_amd.meta_batch import meta_features
    with torch.no_grad():
        f0, f1 = meta_features(model.eval(), mb), meta_features(model2.eval(), mb)
    for a, b in zip(f0, f1):
        assert torch.equal(a, b)
    assert torch.equal(phi0, phi1)
    for a, b in ((preds0, preds1), (var0, var1)):
        assert torch.equal(torch.as_tensor(a), torch.as_tensor(b))
    assert ((preds0 >= 0) & (preds0 <= 1)).all()
