"""Meta-batches of few-shot tasks for the deep-kernel model: the data layout on either side of the hot path.

The reference feeds ONE task at a time (``DKTBatch``, fs_mol/data/dkt.py:32-46: support / query ``MoleculeDKTFeatures``
= an ``FSMolBatch`` disconnected graph + fingerprints + descriptors, fs_mol/data/fsmol_batcher.py:22-54) and runs the
feature extractor >= 3 times forward and h+1 times backward per task.  Here all molecules of all tasks of a meta-batch
form ONE disconnected graph, the extractor runs once, and index maps scatter the molecule features into the padded
``[T, N_max, d]`` layout the HIP library consumes (ragged sizes through ``n_s`` / ``n_q``).
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import List, Optional, Sequence

import torch

from .gnn import GraphBatch, concat_graph_batches


@dataclass
class MoleculeFeatures:
    """``MoleculeDKTFeatures`` (fs_mol/data/dkt.py:25-29) as torch tensors."""

    node_features: torch.Tensor
    adjacency_lists: List[torch.Tensor]
    node_to_graph: torch.Tensor
    num_graphs: int
    fingerprints: torch.Tensor   # [G, 2048]
    descriptors: torch.Tensor    # [G, 42]
    plan: object = None          # graph constants of the extractor, built once per batch (gnn.GraphBatch.with_plan)

    def graph(self) -> GraphBatch:
        return GraphBatch(self.node_features, self.adjacency_lists, self.node_to_graph, self.num_graphs, self.plan)

    def to(self, device):
        g = self.graph().to(device)
        return MoleculeFeatures(g.node_features, g.adjacency_lists, g.node_to_graph, g.num_graphs,
                                self.fingerprints.to(device), self.descriptors.to(device), g.plan)


@dataclass
class DKTBatch:
    """Same fields as the reference's ``DKTBatch`` (fs_mol/data/dkt.py:32-46)."""

    support_features: MoleculeFeatures
    support_labels: torch.Tensor
    support_numeric_labels: torch.Tensor
    query_features: MoleculeFeatures
    query_labels: torch.Tensor
    query_numeric_labels: torch.Tensor

    @property
    def num_support_samples(self) -> int:
        return self.support_features.num_graphs

    @property
    def num_query_samples(self) -> int:
        return self.query_features.num_graphs

    def to(self, device):
        return DKTBatch(self.support_features.to(device), self.support_labels.to(device), self.support_numeric_labels.to(device),
                        self.query_features.to(device), self.query_labels.to(device), self.query_numeric_labels.to(device))


def _as_tensor(x, dtype=None) -> torch.Tensor:
    t = x if isinstance(x, torch.Tensor) else torch.as_tensor(x)      # numpy arrays, lists, tensors alike
    return t if dtype is None else t.to(dtype)


def molecule_features_from_fsmol(part) -> MoleculeFeatures:
    """The reference's ``MoleculeDKTFeatures`` (fs_mol/data/dkt.py:25-29, an ``FSMolBatch`` of NUMPY arrays,
    fs_mol/data/fsmol_batcher.py:22-54: ``node_features [V, 32] float``, ``adjacency_lists`` = one ``[E_t, 2]`` integer array
    per edge type, ``node_to_graph [V]``, ``num_graphs``; plus ``fingerprints [G, 2048]``, ``descriptors [G, 42]``) or its
    ``torchify``-ed twin (fs_mol/utils/torch_utils.py:7-28) -> ``MoleculeFeatures``.  Duck-typed: nothing of the reference is
    imported; ``edge_features`` / ``num_nodes`` / ``num_edges`` are ignored, as the reference's GNN ignores them
    (fs_mol/modules/graph_feature_extractor.py:76-98)."""
    adj = [_as_tensor(a, torch.long).reshape(-1, 2) for a in part.adjacency_lists]
    nodes = _as_tensor(part.node_features, torch.float32)
    n2g = _as_tensor(part.node_to_graph, torch.long)
    G = int(part.num_graphs)
    if nodes.dim() != 2 or n2g.shape[0] != nodes.shape[0]:
        raise ValueError(f"node_features {tuple(nodes.shape)} and node_to_graph {tuple(n2g.shape)} do not describe the same nodes")
    if n2g.numel() and (int(n2g.min()) < 0 or int(n2g.max()) >= G):
        raise ValueError("node_to_graph refers to a graph outside [0, num_graphs)")
    for a in adj:
        if a.numel() and (int(a.min()) < 0 or int(a.max()) >= nodes.shape[0]):
            raise ValueError("an adjacency list refers to a node outside [0, V)")
    fp = _as_tensor(part.fingerprints, torch.float32)       # count fingerprints: integers in the reference's files
    ds = _as_tensor(part.descriptors, torch.float32)
    if fp.shape[0] != G or ds.shape[0] != G:
        raise ValueError(f"fingerprints / descriptors have {fp.shape[0]} / {ds.shape[0]} rows for {G} graphs")
    return MoleculeFeatures(nodes, adj, n2g, G, fp, ds)


def dkt_batch_from_fsmol(batch) -> DKTBatch:
    """The reference's ``DKTBatch`` (fs_mol/data/dkt.py:32-46; numpy or torchified) -> this package's ``DKTBatch``: what
    ``ADKTModel.forward`` and ``collate_meta_batch`` consume.  Labels keep the reference's conventions (bool activity labels,
    numeric labels already log-standardised by ``task_sample_to_dkt_task_sample``, fs_mol/data/dkt.py:86-97)."""
    return DKTBatch(molecule_features_from_fsmol(batch.support_features), _as_tensor(batch.support_labels).bool(),
                    _as_tensor(batch.support_numeric_labels, torch.float32),
                    molecule_features_from_fsmol(batch.query_features), _as_tensor(batch.query_labels).bool(),
                    _as_tensor(batch.query_numeric_labels, torch.float32))


def _concat_molecules(parts: Sequence[MoleculeFeatures]) -> MoleculeFeatures:
    g = concat_graph_batches([p.graph() for p in parts])
    return MoleculeFeatures(g.node_features, g.adjacency_lists, g.node_to_graph, g.num_graphs,
                            torch.cat([p.fingerprints for p in parts]), torch.cat([p.descriptors for p in parts]))


@dataclass
class MetaBatch:
    molecules: MoleculeFeatures   # every molecule of every task, one disconnected graph
    s_index: torch.Tensor         # [T, Ns_max] molecule id of each support slot (0 where padded)
    q_index: torch.Tensor         # [T, Nq_max]
    s_mask: torch.Tensor          # [T, Ns_max] 1.0 for real slots
    q_mask: torch.Tensor
    n_s: torch.Tensor             # [T] int32
    n_q: torch.Tensor
    support_labels: torch.Tensor          # [T, Ns_max] bool (False where padded)
    support_numeric_labels: torch.Tensor  # [T, Ns_max]
    query_labels: torch.Tensor
    query_numeric_labels: torch.Tensor

    @property
    def num_tasks(self) -> int:
        return self.s_index.shape[0]

    def to(self, device):
        return MetaBatch(self.molecules.to(device), *(t.to(device) for t in (
            self.s_index, self.q_index, self.s_mask, self.q_mask, self.n_s, self.n_q, self.support_labels,
            self.support_numeric_labels, self.query_labels, self.query_numeric_labels)))

    def labels(self, use_numeric_labels: bool):
        """±1 for bool labels (fs_mol/models/adaptive_dkt.py:207-209) or the (already standardised) numeric ones."""
        if use_numeric_labels:
            return self.support_numeric_labels.float(), self.query_numeric_labels.float()
        cv = lambda l: (l.float() - 0.5) * 2.0
        return cv(self.support_labels) * self.s_mask, cv(self.query_labels) * self.q_mask


def collate_meta_batch(tasks: Sequence[DKTBatch], pad_to: int = 4) -> MetaBatch:
    """Many single-task batches -> one meta-batch.  Padded sizes are rounded up to a multiple of ``pad_to`` so that
    the library's 16-byte vector loads stay legal."""
    T = len(tasks)
    ns = [t.num_support_samples for t in tasks]
    nq = [t.num_query_samples for t in tasks]
    up = lambda x: ((max(x) + pad_to - 1) // pad_to) * pad_to
    Ns, Nq = up(ns), up(nq)
    parts, s_index, q_index = [], torch.zeros(T, Ns, dtype=torch.long), torch.zeros(T, Nq, dtype=torch.long)
    s_mask, q_mask = torch.zeros(T, Ns), torch.zeros(T, Nq)
    sl, snl = torch.zeros(T, Ns, dtype=torch.bool), torch.zeros(T, Ns)
    ql, qnl = torch.zeros(T, Nq, dtype=torch.bool), torch.zeros(T, Nq)
    g0 = 0
    for t, b in enumerate(tasks):
        parts += [b.support_features, b.query_features]
        s_index[t, :ns[t]] = torch.arange(g0, g0 + ns[t]); g0 += ns[t]
        q_index[t, :nq[t]] = torch.arange(g0, g0 + nq[t]); g0 += nq[t]
        s_mask[t, :ns[t]] = 1.0
        q_mask[t, :nq[t]] = 1.0
        sl[t, :ns[t]] = b.support_labels.bool().cpu()
        ql[t, :nq[t]] = b.query_labels.bool().cpu()
        snl[t, :ns[t]] = b.support_numeric_labels.float().cpu()
        qnl[t, :nq[t]] = b.query_numeric_labels.float().cpu()
    mols = _concat_molecules([p.to("cpu") for p in parts])
    # the extractor's graph constants (edge lists both ways, degrees, CSR lists of the fused kernels): once per batch, here on the
    # host, instead of ~25 launches and three host synchronisations per forward on the device
    mols.plan = mols.graph().with_plan().plan
    return MetaBatch(mols, s_index, q_index, s_mask, q_mask, torch.tensor(ns, dtype=torch.int32),
                     torch.tensor(nq, dtype=torch.int32), sl, snl, ql, qnl)


def shard_tasks_by_nodes(tasks: Sequence[DKTBatch], world: int, rank: int) -> List[int]:
    """Indices of the tasks rank ``rank`` of ``world`` owns, balanced by GRAPH NODE count rather than task count
    (with the GNN on the path a task costs what its molecules cost, and query sets range from 16 to 256
    molecules).  Longest-processing-time greedy: heaviest task first onto the lightest rank; ties break on the lower
    index, so every rank computes the same assignment from the same task list without talking to the others.  Ranks
    may end up with different task counts: run the step with ``MetaStepConfig(uneven_shards=True)``."""
    cost = [int(t.support_features.node_features.shape[0]) + int(t.query_features.node_features.shape[0]) for t in tasks]
    order = sorted(range(len(tasks)), key=lambda i: (-cost[i], i))
    load, owner = [0] * world, [0] * len(tasks)
    for i in order:
        r = min(range(world), key=lambda k: (load[k], k))
        owner[i] = r
        load[r] += cost[i]
    return [i for i in range(len(tasks)) if owner[i] == rank]


def _batch_coupled(model) -> Optional[str]:
    """Why ONE forward over all molecules of all tasks would NOT equal the reference's per-task support / query forwards."""
    gfe = getattr(model, "graph_feature_extractor", None)
    if gfe is None or not model.training:
        return None
    if gfe.config.output_norm == "batch":
        return "output_norm='batch' (train-mode BatchNorm statistics would be pooled over tasks and over support + query sets)"
    if gfe.config.gnn_config.dropout_rate > 0.0:
        return "dropout_rate > 0 (one mask draw per molecule set in the reference, fs_mol/modules/gnn.py:497-513)"
    return None


def meta_features(model, mb: MetaBatch):
    """ONE forward of the deep-kernel feature extractor for the whole meta-batch ->
    ``Z_s [T, Ns_max, d]``, ``Z_q [T, Nq_max, d]`` (padded rows are zero and receive zero gradient).
    Exact only when molecules do not interact inside the extractor: a model whose extractor couples the molecules of a
    forward pass in train mode (BatchNorm output norm, dropout) is refused - the reference runs separate support and query
    forwards per task (fs_mol/models/adaptive_dkt.py:141-160) and pooling would silently change the features."""
    why = _batch_coupled(model)
    if why is not None:
        raise NotImplementedError("meta_features: the batched extractor forward is not equivalent to the reference's per-task "
                                  "forwards with " + why + "; use the per-task path (ADKTModel.forward + cauchy_hypergradient)")
    F = model._features(mb.molecules)  # [G_total, d]
    Z_s = F[mb.s_index] * mb.s_mask.unsqueeze(-1).to(F.dtype)
    Z_q = F[mb.q_index] * mb.q_mask.unsqueeze(-1).to(F.dtype)
    return Z_s, Z_q


def model_meta_step(model, optimizer, mb: MetaBatch, cfg=None, distributed: bool = False, check: bool = False,
                    uneven_shards: bool = False, lr_scheduler=None):
    """The batched counterpart of one iteration of ``ADKTModelTrainer.train_loop``
    (fs_mol/utils/adaptive_dkt_utils.py:352-413) for an ``ADKTModel``: returns per-task per-sample losses.
    ``lr_scheduler`` (e.g. the warm-up of ``checkpoint.load_model_gnn_weights``) is stepped after the optimiser (:412-413)."""
    from .trainer import MetaStepConfig, meta_step

    if cfg is None:
        c = model.config
        cfg = MetaStepConfig(gp_kernel=c.gp_kernel, use_numeric_labels=c.use_numeric_labels, use_ard=c.use_ard,
                             use_lengthscale_prior=c.use_lengthscale_prior, ignore_grad_correction=c.ignore_grad_correction,
                             uneven_shards=uneven_shards)
    y_s, y_q = mb.labels(cfg.use_numeric_labels)
    out = meta_step(lambda: meta_features(model, mb), list(model.feature_extractor_params()), optimizer, y_s, y_q, cfg,
                    n_s=mb.n_s, n_q=mb.n_q, distributed=distributed, check=check)
    if lr_scheduler is not None:
        lr_scheduler.step()
    return out
