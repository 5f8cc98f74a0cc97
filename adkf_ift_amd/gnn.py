"""GNN feature extractor, re-authored for PyTorch-ROCm with native index/scatter ops.

Same function as the reference's ``GraphFeatureExtractor`` (fs_mol/modules/graph_feature_extractor.py:43-98):
``init_node_proj`` -> ``num_layers`` x ``GNNBlock`` (PNA multi-aggregation relational message passing in
``num_heads`` "towers", ReZero, BOOM MLP; fs_mol/modules/gnn.py:168-265, 389-515) -> ``CombinedGraphReadout``
(weighted-mean + weighted-sum + max; fs_mol/modules/graph_readout.py:119-296), default sizes of Appendix B of the
survey.  What is different is the execution plan, chosen for a GPU that wants few, fat kernels:

* the reference runs ``towers x edge_types`` tiny ``Linear(2*32 -> 3*64)`` modules per block and gathers node states
  once per tower and edge type; here each edge type gathers the full node state ONCE and all towers go through one
  batched GEMM (``einsum('ehi,hio->eho')`` on a ``[towers, 2*in, 3*msg]`` weight);
* everything that depends only on the graph (concatenated targets, in-degrees, PNA scalers, bidirectional edge
  lists) is computed once per forward, not once per tower per layer;
* the four first-layer MLPs of the two weighted read-outs are one GEMM over the ``[V, 1408]`` node states;
* ``torch_scatter`` (not installable here) is replaced by ``index_add_`` / ``scatter_reduce_`` whose empty-segment
  conventions are made to match (``scatter_max`` / ``scatter_mean`` of an empty segment = 0);
* any number of tasks is concatenated into ONE disconnected graph (``concat_graph_batches``), so a meta-batch costs
  one forward and one backward of the extractor instead of >= 3 + (h+1) per task.

``mp_norm_layer`` exists in the reference block but is never applied in its forward (gnn.py:477-515); it is kept as
an (unused) submodule only so that reference checkpoints map one-to-one (``load_reference_state_dict``).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import dense

SMALL_NUMBER = 1e-7
_FUSED_BLOCK = __import__("os").environ.get("ADKF_GNN_FUSED_BLOCK", "1") != "0"   # diagnostics: 0 keeps the PyTorch ops in the middle of a block
_FUSED_MP = __import__("os").environ.get("ADKF_GNN_FUSED_MP", "1") != "0"   # diagnostics: 0 keeps message functions and aggregation as two autograd nodes
_POOL_HIDDEN = __import__("os").environ.get("ADKF_READOUT_POOL_HIDDEN", "1") != "0"   # diagnostics: 0 pools the value MLPs' outputs (round 3's order)
NUM_NODE_FEATURES = 32   # fs_mol/data/fsmol_dataset.py:21
NUM_EDGE_TYPES = 3       # fs_mol/data/fsmol_dataset.py:22
PNA_DELTA = 1.1515       # fs_mol/modules/gnn.py:237


@dataclass
class GNNConfig:
    """CLI defaults of the reference (fs_mol/modules/gnn.py:31-63), not its dataclass defaults."""

    type: str = "PNA"
    num_edge_types: int = NUM_EDGE_TYPES
    hidden_dim: int = 128
    num_heads: int = 4
    per_head_dim: int = 64
    intermediate_dim: int = 1024
    message_function_depth: int = 1
    num_layers: int = 10
    dropout_rate: float = 0.0
    use_rezero_scaling: bool = True
    make_edges_bidirectional: bool = True


@dataclass
class GraphReadoutConfig:
    readout_type: str = "combined"
    use_all_states: bool = True
    num_heads: int = 12
    head_dim: int = 64
    output_dim: int = 512


@dataclass
class GraphFeatureExtractorConfig:
    initial_node_feature_dim: int = NUM_NODE_FEATURES
    gnn_config: GNNConfig = field(default_factory=GNNConfig)
    readout_config: GraphReadoutConfig = field(default_factory=GraphReadoutConfig)
    output_norm: str = "off"


@dataclass
class GraphBatch:
    """The fields of ``FSMolBatch`` (fs_mol/data/fsmol_batcher.py:22-54) the extractor reads, as torch tensors."""

    node_features: torch.Tensor            # [V, F] float
    adjacency_lists: List[torch.Tensor]    # num_edge_types x [E_t, 2] int64 (src, tgt)
    node_to_graph: torch.Tensor            # [V] int64
    num_graphs: int
    plan: Optional["_GraphPlan"] = None    # graph constants built once per batch (with_plan); None: built per forward

    def to(self, device):
        return GraphBatch(self.node_features.to(device), [a.to(device) for a in self.adjacency_lists],
                          self.node_to_graph.to(device), self.num_graphs, self.plan.to(device) if self.plan is not None else None)

    def with_plan(self, bidirectional: bool = True, pna: bool = True, dtype: torch.dtype = torch.float32) -> "GraphBatch":
        """Builds the graph constants here (on whatever device the batch lives on - the host, in a data pipeline) so that the
        extractor's forward does not: ``batch.with_plan().to(device)``."""
        self.plan = _GraphPlan(self.adjacency_lists, self.node_features.shape[0], bidirectional, pna, dtype,
                               self.node_to_graph, self.num_graphs)
        return self


def concat_graph_batches(batches: Sequence[GraphBatch]) -> GraphBatch:
    """Many disconnected-graph batches -> one (node and graph ids renumbered)."""
    nodes, n2g, adj = [], [], [[] for _ in batches[0].adjacency_lists]
    v0 = g0 = 0
    for b in batches:
        nodes.append(b.node_features)
        n2g.append(b.node_to_graph + g0)
        for t, a in enumerate(b.adjacency_lists):
            adj[t].append(a + v0)
        v0 += b.node_features.shape[0]
        g0 += b.num_graphs
    return GraphBatch(torch.cat(nodes), [torch.cat(a) for a in adj], torch.cat(n2g), g0)


class _GraphPlan:
    """Graph constants shared by all layers and towers: bidirectional edge lists, degrees and PNA scalers, the CSR lists of the
    fused kernels (by target, by source, nodes by graph).  They depend on the batch alone, so a data pipeline builds them ONCE per
    batch on the host (``GraphBatch.with_plan`` - ``collate_meta_batch`` does) and ships them with it: built on the device per
    forward they cost ~25 tiny launches and three host synchronisations (``torch.bincount`` returns a size)."""

    def to(self, device):
        new = object.__new__(_GraphPlan)
        for k, v in self.__dict__.items():
            if isinstance(v, torch.Tensor):
                v = v.to(device)
            elif isinstance(v, list):
                v = [t.to(device) for t in v]
            setattr(new, k, v)
        return new

    def matches(self, num_nodes: int, bidirectional: bool, pna: bool, device, dtype, adjacency_lists=None) -> bool:
        """``adjacency_lists`` (when given): the per-edge-type edge counts must be the plan's - a batch whose lists were replaced or
        filtered after collation must not run on the old CSR lists and degrees."""
        if adjacency_lists is not None:
            want = [int(a.shape[0]) * (2 if self.bidirectional else 1) for a in adjacency_lists]
            if want != [int(s.shape[0]) for s in self.srcs]:
                return False
        return (self.num_nodes == num_nodes and self.bidirectional == bidirectional and (self.pna or not pna)
                and self.all_tgts.device == device and self.inv_count.dtype == dtype)

    def __init__(self, adjacency_lists: List[torch.Tensor], num_nodes: int, bidirectional: bool, pna: bool,
                 dtype: torch.dtype = torch.float32, node_to_graph: Optional[torch.Tensor] = None, num_graphs: int = 0):
        self.bidirectional, self.pna = bidirectional, pna
        if bidirectional:  # fs_mol/modules/gnn.py:540-544
            adjacency_lists = [torch.cat((a, a.flip(1)), dim=0) for a in adjacency_lists]
        self.srcs = [a[:, 0].contiguous() for a in adjacency_lists]   # contiguous: the HIP kernels index them directly
        self.tgts = [a[:, 1].contiguous() for a in adjacency_lists]
        self.all_tgts = torch.cat(self.tgts) if self.tgts else torch.zeros(0, dtype=torch.long)
        self.num_nodes = num_nodes
        counts = torch.bincount(self.all_tgts, minlength=num_nodes)
        deg = counts.to(dtype)
        self.inv_count = 1.0 / deg.clamp(min=1.0)                      # scatter_mean: empty segment -> 0
        # segments of the concatenated message list by target node, for the fused aggregation kernel (csrc/pna.h)
        self.perm = torch.argsort(self.all_tgts, stable=True)
        self.rowptr = torch.cat((counts.new_zeros(1), torch.cumsum(counts, 0)))
        # ... and by SOURCE node: the backward of the message functions gathers d x over each node's outgoing and incoming
        # edges in these two fixed orders instead of scatter-adding atomically (bit-reproducible gradients)
        all_srcs = torch.cat(self.srcs) if self.srcs else torch.zeros(0, dtype=torch.long)
        out_counts = torch.bincount(all_srcs, minlength=num_nodes)
        self.perm_src = torch.argsort(all_srcs, stable=True)
        self.rowptr_src = torch.cat((out_counts.new_zeros(1), torch.cumsum(out_counts, 0)))
        if pna:
            log_deg = torch.log(deg + 1.0)
            self.amplify = (log_deg / PNA_DELTA).unsqueeze(-1)                      # gnn.py:241
            self.attenuate = (PNA_DELTA / (log_deg + SMALL_NUMBER)).unsqueeze(-1)   # gnn.py:242
        # nodes by graph, for the read-out pooling kernels (csrc/readout.h)
        self.num_graphs = int(num_graphs)
        self.perm_graph = self.rowptr_graph = None
        if node_to_graph is not None:
            gcounts = torch.bincount(node_to_graph, minlength=self.num_graphs)
            self.perm_graph = torch.argsort(node_to_graph, stable=True)
            self.rowptr_graph = torch.cat((gcounts.new_zeros(1), torch.cumsum(gcounts, 0)))


class TowerMessagePassing(nn.Module):
    """All ``num_heads`` towers of one block at once: relational message functions per edge type
    (fs_mol/modules/gnn.py:95-148) and PNA / multi-aggregation / plain aggregation (:150-265)."""

    def __init__(self, config: GNNConfig):
        super().__init__()
        self.kind = config.type.lower()
        if self.kind not in ("pna", "multiaggr", "plain"):
            raise ValueError(f"Unknown GNN type {config.type}.")  # MultiHeadAttention: other ablation, out of scope
        H, self.in_dim = config.num_heads, config.hidden_dim // config.num_heads
        assert config.hidden_dim % config.num_heads == 0, "Number of heads needs to divide GNN hidden dim."
        self.H, self.msg = H, config.per_head_dim
        self.out_msg = (3 if self.kind != "plain" else 1) * self.msg
        dims = [2 * self.in_dim] * config.message_function_depth + [self.out_msg]
        self.weights = nn.ParameterList()   # [edge_type][layer] -> [H, in, out]
        self.biases = nn.ParameterList()
        self.depth = config.message_function_depth
        self.capture: Optional[list] = None   # diagnostics: a list here receives the post-ReLU messages [E_all, H, out] of each forward
        self.capture_argmax: Optional[list] = None   # ... and here the arg-max message ids [V, H, m] of the max aggregation (GPU path)
        for _ in range(config.num_edge_types):
            for l in range(self.depth):
                w = torch.empty(H, dims[l], dims[l + 1])
                b = torch.empty(H, dims[l + 1])
                for h in range(H):  # nn.Linear's default init, per tower
                    nn.init.kaiming_uniform_(w[h].t(), a=math.sqrt(5))
                    bound = 1.0 / math.sqrt(dims[l])
                    nn.init.uniform_(b[h], -bound, bound)
                self.weights.append(nn.Parameter(w))
                self.biases.append(nn.Parameter(b))

    @property
    def message_size(self) -> int:
        per_tower = {"plain": self.msg, "multiaggr": 4 * self.msg, "pna": 12 * self.msg}[self.kind]
        return self.H * per_tower

    def forward(self, x: torch.Tensor, plan: _GraphPlan, scale: bool = True) -> torch.Tensor:
        """scale=False (PNA only): return the UNSCALED aggregates [V, H * 4m]; the caller folds the identity / amplify /
        attenuate scalers into the output projection (GNNBlock) instead of materialising the [V, H * 12m] concatenation."""
        V, H, m = x.shape[0], self.H, self.msg
        xt = x.view(V, H, self.in_dim)
        if x.is_cuda and x.dtype == torch.float32 and self.depth == 1 and self.kind != "plain":
            # GPU fast path (csrc/pna.h): per edge type ONE batched MFMA GEMM that gathers source / target states on the fly
            # and applies bias + ReLU in its epilogue, then ONE aggregation kernel; no fallback here - a missing library raises
            if _FUSED_MP and self.out_msg == 3 * m:
                agg, amax, msgs = _MessagePass.apply(x.contiguous(), plan, H, self.in_dim, self.out_msg, *self.weights, *self.biases)
            else:
                msgs = _MessageFunction.apply(x.contiguous(), plan, H, self.in_dim, self.out_msg, *self.weights, *self.biases)
                agg, amax = _PNAAggregate.apply(msgs, plan.perm, plan.rowptr, V)
            if self.capture is not None:
                self.capture.append(msgs.detach())
            if self.capture_argmax is not None:
                self.capture_argmax.append(amax)
            if self.kind == "pna" and scale:
                amp, att = plan.amplify.unsqueeze(-1).to(x.dtype), plan.attenuate.unsqueeze(-1).to(x.dtype)
                agg = torch.cat((agg, amp * agg, att * agg), dim=2)
            return agg.reshape(V, -1)
        msgs = []
        for et in range(len(plan.srcs)):
            h = torch.cat((xt[plan.srcs[et]], xt[plan.tgts[et]]), dim=2)       # [E, H, 2 in]
            for l in range(self.depth):
                k = et * self.depth + l
                h = torch.einsum("ehi,hio->eho", h, self.weights[k]) + self.biases[k]
                if l + 1 < self.depth:
                    h = F.relu(h)
            msgs.append(F.relu(h))                                             # gnn.py:141
        msgs = torch.cat(msgs, dim=0)                                          # [E_all, H, out_msg]
        if self.capture is not None:
            self.capture.append(msgs.detach())
        tg = plan.all_tgts
        if self.kind == "plain":
            return x.new_zeros(V, H, m).index_add_(0, tg, msgs).reshape(V, -1)
        if msgs.is_cuda and msgs.dtype == torch.float32:
            # one HIP kernel for sum | mean | std | max (and one for their backward) instead of the ~15 (~30) element-wise,
            # index and scatter launches below per layer; no fallback on the GPU: a missing library raises
            agg, _ = _PNAAggregate.apply(msgs.contiguous(), plan.perm, plan.rowptr, V)
            if self.kind == "pna":
                amp, att = plan.amplify.unsqueeze(-1).to(x.dtype), plan.attenuate.unsqueeze(-1).to(x.dtype)
                agg = torch.cat((agg, amp * agg, att * agg), dim=2)
            return agg.reshape(V, -1)
        s_sum = x.new_zeros(V, H, m).index_add_(0, tg, msgs[..., :m])
        mean_msgs = msgs[..., m:2 * m]
        s_mean = x.new_zeros(V, H, m).index_add_(0, tg, mean_msgs) * plan.inv_count.view(V, 1, 1).to(x.dtype)
        dev = F.relu(mean_msgs.pow(2) - s_mean[tg].pow(2)) + SMALL_NUMBER      # gnn.py:213-216
        s_std = torch.sqrt(x.new_zeros(V, H, m).index_add_(0, tg, dev))
        idx = tg.view(-1, 1, 1).expand(-1, H, m)
        s_max = x.new_zeros(V, H, m).scatter_reduce_(0, idx, msgs[..., 2 * m:3 * m], reduce="amax", include_self=False)
        agg = torch.cat((s_sum, s_mean, s_std, s_max), dim=2)                  # [V, H, 4m], tower-major like the reference cat
        if self.kind == "pna" and scale:
            amp, att = plan.amplify.unsqueeze(-1).to(x.dtype), plan.attenuate.unsqueeze(-1).to(x.dtype)
            agg = torch.cat((agg, amp * agg, att * agg), dim=2)                # gnn.py:244-251
        return agg.reshape(V, -1)


class _MessageFunction(torch.autograd.Function):
    """relu(cat(x[src], x[tgt]) W_et + b_et) for every edge type and tower -> [E_all, H, out] (``adkf_msg_forward`` /
    ``adkf_msg_backward``, csrc/pna.h: ONE launch of each kind for all edge types).  x [V, H*in] float32 contiguous; weights[et]
    [H, 2 in, out], biases[et] [H, out]."""

    @staticmethod
    def _table(plan, weights, biases=None, dWs=None, dbs=None):
        import ctypes as C

        from . import _lib
        n_et = len(weights)
        tab = (_lib.MsgEt * n_et)()
        for et in range(n_et):
            tab[et].src, tab[et].tgt = plan.srcs[et].data_ptr(), plan.tgts[et].data_ptr()
            tab[et].W = weights[et].data_ptr()
            tab[et].bias = biases[et].data_ptr() if biases is not None else None
            tab[et].dW = dWs[et].data_ptr() if dWs is not None else None
            tab[et].db = dbs[et].data_ptr() if dbs is not None else None
            tab[et].E = int(plan.srcs[et].shape[0])
        return tab

    @staticmethod
    def forward(ctx, x, plan, H, in_dim, out_dim, *params):
        import ctypes as C

        from . import _lib
        lib = _lib.load()
        n_et = len(params) // 2
        weights, biases = [w.contiguous() for w in params[:n_et]], [b.contiguous() for b in params[n_et:]]
        E_all = int(plan.all_tgts.shape[0])
        msgs = torch.empty(E_all, H, out_dim, dtype=torch.float32, device=x.device)
        st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        tab = _MessageFunction._table(plan, weights, biases)
        _lib.check(lib.adkf_msg_forward(C.c_void_p(x.data_ptr()), C.cast(tab, C.c_void_p), n_et, H, in_dim, out_dim,
                                        C.c_void_p(msgs.data_ptr()), st), "adkf_msg_forward")
        ctx.save_for_backward(x, msgs, *weights)
        ctx.plan, ctx.dims, ctx.n_et = plan, (H, in_dim, out_dim), n_et
        return msgs

    @staticmethod
    def backward(ctx, d_msgs):
        import ctypes as C

        from . import _lib
        lib = _lib.load()
        x, msgs, *weights = ctx.saved_tensors
        plan, (H, in_dim, out_dim), n_et = ctx.plan, ctx.dims, ctx.n_et
        d_msgs = d_msgs.contiguous()
        dev = x.device
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        E_all = int(plan.all_tgts.shape[0])
        # no floating-point atomics anywhere (csrc/pna.h): d cat is written once per edge and d x gathered over each node's
        # edge lists; d W / d b are per-chunk partials summed in a fixed order - every output element is written, none pre-filled
        dcat = torch.empty(max(E_all, 1), H, 2 * in_dim, dtype=torch.float32, device=dev)
        dW_all = [torch.empty_like(w) for w in weights]
        db_all = torch.empty(n_et, H, out_dim, dtype=torch.float32, device=dev)
        dbs = [db_all[et] for et in range(n_et)]
        tab = _MessageFunction._table(plan, weights, None, dW_all, dbs)
        need = int(lib.adkf_msg_backward_scratch_bytes(C.cast(tab, C.c_void_p), n_et, H, in_dim, out_dim))
        scratch = torch.empty(max(need, 4) // 4, dtype=torch.float32, device=dev)
        dx = torch.empty_like(x)
        _lib.check(lib.adkf_msg_backward(ptr(x), C.cast(tab, C.c_void_p), n_et, H, in_dim, out_dim, ptr(msgs), ptr(d_msgs),
                                         ptr(plan.perm_src), ptr(plan.rowptr_src), ptr(plan.perm), ptr(plan.rowptr), x.shape[0],
                                         ptr(dcat), ptr(dx), ptr(scratch), scratch.numel() * 4, st), "adkf_msg_backward")
        return (dx, None, None, None, None, *dW_all, *dbs)


class _PNAAggregate(torch.autograd.Function):
    """[E, H, 3m] messages -> [V, H, 4m] (sum | mean | std | max) through ``adkf_pna_aggregate`` (csrc/pna.h)."""

    @staticmethod
    def forward(ctx, msgs, perm, rowptr, V):
        import ctypes as C

        from . import _lib
        lib = _lib.load()
        E, H, m3 = msgs.shape
        m = m3 // 3
        agg = torch.empty(V, H, 4 * m, dtype=torch.float32, device=msgs.device)
        argmax = torch.empty(V, H, m, dtype=torch.int32, device=msgs.device)
        st = C.c_void_p(torch.cuda.current_stream(msgs.device).cuda_stream)
        _lib.check(lib.adkf_pna_aggregate(C.c_void_p(msgs.data_ptr()), C.c_void_p(perm.data_ptr()), C.c_void_p(rowptr.data_ptr()),
                                          V, H, m, C.c_void_p(agg.data_ptr()), C.c_void_p(argmax.data_ptr()), st), "adkf_pna_aggregate")
        ctx.save_for_backward(msgs, perm, rowptr, agg, argmax)
        ctx.mark_non_differentiable(argmax)
        return agg, argmax

    @staticmethod
    def backward(ctx, d_agg, _d_argmax=None):
        import ctypes as C

        from . import _lib
        lib = _lib.load()
        msgs, perm, rowptr, agg, argmax = ctx.saved_tensors
        V, H, m4 = agg.shape
        d_agg = d_agg.contiguous()
        d_msgs = torch.empty_like(msgs)
        st = C.c_void_p(torch.cuda.current_stream(msgs.device).cuda_stream)
        _lib.check(lib.adkf_pna_aggregate_backward(C.c_void_p(msgs.data_ptr()), C.c_void_p(perm.data_ptr()), C.c_void_p(rowptr.data_ptr()),
                                                   C.c_void_p(agg.data_ptr()), C.c_void_p(argmax.data_ptr()), C.c_void_p(d_agg.data_ptr()),
                                                   V, H, m4 // 4, C.c_void_p(d_msgs.data_ptr()), st), "adkf_pna_aggregate_backward")
        return d_msgs, None, None, None


class _MessagePass(torch.autograd.Function):
    """``_MessageFunction`` followed by ``_PNAAggregate`` as ONE node of the graph: x -> (agg [V, H, 4m], argmax, msgs).  Same
    kernels forward; in the backward the aggregation kernel writes the gradient in front of the messages' ReLU
    (``adkf_pna_aggregate_backward_relu``) and the three products of ``adkf_msg_backward`` read that one tensor instead of
    gradient + mask - they are bound by exactly that traffic (csrc/pna.h).  ``msgs`` is returned for diagnostics only."""

    @staticmethod
    def forward(ctx, x, plan, H, in_dim, out_dim, *params):
        import ctypes as C

        from . import _lib
        lib = _lib.load()
        n_et = len(params) // 2
        weights, biases = [w.contiguous() for w in params[:n_et]], [b.contiguous() for b in params[n_et:]]
        E_all, V, m = int(plan.all_tgts.shape[0]), x.shape[0], out_dim // 3
        dev = x.device
        msgs = torch.empty(E_all, H, out_dim, dtype=torch.float32, device=dev)
        agg = torch.empty(V, H, 4 * m, dtype=torch.float32, device=dev)
        argmax = torch.empty(V, H, m, dtype=torch.int32, device=dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        tab = _MessageFunction._table(plan, weights, biases)
        _lib.check(lib.adkf_msg_forward(ptr(x), C.cast(tab, C.c_void_p), n_et, H, in_dim, out_dim, ptr(msgs), st), "adkf_msg_forward")
        _lib.check(lib.adkf_pna_aggregate(ptr(msgs), ptr(plan.perm), ptr(plan.rowptr), V, H, m, ptr(agg), ptr(argmax), st), "adkf_pna_aggregate")
        ctx.save_for_backward(x, msgs, agg, argmax, *weights)
        ctx.plan, ctx.dims, ctx.n_et = plan, (H, in_dim, out_dim), n_et
        ctx.mark_non_differentiable(argmax, msgs)
        return agg, argmax, msgs

    @staticmethod
    def backward(ctx, d_agg, _d_argmax=None, _d_msgs=None):
        import ctypes as C

        from . import _lib
        lib = _lib.load()
        x, msgs, agg, argmax, *weights = ctx.saved_tensors
        plan, (H, in_dim, out_dim), n_et = ctx.plan, ctx.dims, ctx.n_et
        dev, V, m = x.device, x.shape[0], out_dim // 3
        d_agg = d_agg.contiguous()
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        E_all = int(plan.all_tgts.shape[0])
        d_pre = torch.empty_like(msgs)
        if E_all > 0:
            _lib.check(lib.adkf_pna_aggregate_backward_relu(ptr(msgs), ptr(plan.perm), ptr(plan.rowptr), ptr(agg), ptr(argmax), ptr(d_agg),
                                                            V, H, m, ptr(d_pre), st), "adkf_pna_aggregate_backward_relu")
        dcat = torch.empty(max(E_all, 1), H, 2 * in_dim, dtype=torch.float32, device=dev)
        dW_all = [torch.empty_like(w) for w in weights]
        db_all = torch.empty(n_et, H, out_dim, dtype=torch.float32, device=dev)
        dbs = [db_all[et] for et in range(n_et)]
        tab = _MessageFunction._table(plan, weights, None, dW_all, dbs)
        need = int(lib.adkf_msg_backward_scratch_bytes(C.cast(tab, C.c_void_p), n_et, H, in_dim, out_dim))
        scratch = torch.empty(max(need, 4) // 4, dtype=torch.float32, device=dev)
        dx = torch.empty_like(x)
        _lib.check(lib.adkf_msg_backward(ptr(x), C.cast(tab, C.c_void_p), n_et, H, in_dim, out_dim, None, ptr(d_pre),
                                         ptr(plan.perm_src), ptr(plan.rowptr_src), ptr(plan.perm), ptr(plan.rowptr), V,
                                         ptr(dcat), ptr(dx), ptr(scratch), scratch.numel() * 4, st), "adkf_msg_backward")
        return (dx, None, None, None, None, *dW_all, *dbs)


class _BlockCombine(torch.autograd.Function):
    """new = p0 + amp p1 + att p2 + bias;  x1 = x + alpha new;  h = LayerNorm(x1)  ->  (x1, h)   as one HIP kernel forward and one
    backward (``adkf_block_combine``, csrc/block.h) instead of nine / ~twenty element-wise and reduction launches per block."""

    @staticmethod
    def forward(ctx, p, x, amp, att, bias, alpha, gamma, beta, eps):
        import ctypes as C

        from . import _lib
        lib = _lib.load()
        V, hid = x.shape
        p, x, amp, att = p.contiguous(), x.contiguous(), amp.contiguous(), att.contiguous()
        bias, gamma, beta = bias.contiguous(), gamma.contiguous(), beta.contiguous()
        x1, h = torch.empty_like(x), torch.empty_like(x)
        mu = torch.empty(V, dtype=torch.float32, device=x.device)
        rstd = torch.empty_like(mu)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        st = C.c_void_p(torch.cuda.current_stream(x.device).cuda_stream)
        _lib.check(lib.adkf_block_combine(ptr(p), ptr(x), ptr(amp), ptr(att), ptr(bias), ptr(alpha), ptr(gamma), ptr(beta), float(eps), V, hid,
                                          ptr(x1), ptr(h), ptr(mu), ptr(rstd), st), "adkf_block_combine")
        ctx.save_for_backward(p, x1, amp, att, bias, alpha, gamma, mu, rstd)
        return x1, h

    @staticmethod
    def backward(ctx, g_x1, g_h):
        import ctypes as C

        from . import _lib
        lib = _lib.load()
        p, x1, amp, att, bias, alpha, gamma, mu, rstd = ctx.saved_tensors
        V, hid = x1.shape
        dev = x1.device
        g_x1 = torch.zeros_like(x1) if g_x1 is None else g_x1.contiguous()
        g_h = torch.zeros_like(x1) if g_h is None else g_h.contiguous()
        d_p, d_x = torch.empty_like(p), torch.empty_like(x1)
        d_bias, d_gamma, d_beta, d_alpha = torch.empty_like(bias), torch.empty_like(gamma), torch.empty_like(gamma), torch.empty_like(alpha)
        need = int(lib.adkf_block_combine_scratch_bytes(V, hid))
        scratch = torch.empty(need // 4, dtype=torch.float32, device=dev)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        _lib.check(lib.adkf_block_combine_backward(ptr(p), ptr(x1), ptr(amp), ptr(att), ptr(bias), ptr(alpha), ptr(gamma), ptr(mu), ptr(rstd),
                                                   ptr(g_x1), ptr(g_h), V, hid, ptr(d_p), ptr(d_x), ptr(d_bias), ptr(d_alpha), ptr(d_gamma),
                                                   ptr(d_beta), ptr(scratch), need, st), "adkf_block_combine_backward")
        return d_p, d_x, None, None, d_bias, d_alpha, d_gamma, d_beta, None


class BOOMLayer(nn.Module):
    def __init__(self, inout_dim: int, intermediate_dim: int, dropout: float):
        super().__init__()
        self.linear1 = nn.Linear(inout_dim, intermediate_dim)
        self.linear2 = nn.Linear(intermediate_dim, inout_dim)
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        return dense.linear(self.dropout(F.leaky_relu(self.linear1(x))), self.linear2.weight, self.linear2.bias)


class GNNBlock(nn.Module):
    """v' = v + alpha * MsgOut(MP(v));  v = v' + alpha * BOOM(LN(v'))   (fs_mol/modules/gnn.py:477-515)."""

    def __init__(self, config: GNNConfig):
        super().__init__()
        self.config = config
        if config.use_rezero_scaling:
            self.alpha = nn.Parameter(torch.full((1,), SMALL_NUMBER))
        self.mp = TowerMessagePassing(config)
        self.msg_out_projection = nn.Linear(self.mp.message_size, config.hidden_dim)
        self.mp_norm_layer = nn.LayerNorm(config.hidden_dim)  # present, never applied - exactly like the reference
        if config.intermediate_dim > 0:
            self.boom_layer: Optional[BOOMLayer] = BOOMLayer(config.hidden_dim, config.intermediate_dim, config.dropout_rate)
            self.boom_norm_layer: Optional[nn.Module] = nn.LayerNorm(config.hidden_dim)
        else:
            self.boom_layer = self.boom_norm_layer = None
        self.dropout_layer = nn.Dropout(config.dropout_rate)

    def forward(self, x: torch.Tensor, plan: _GraphPlan) -> torch.Tensor:
        if self.mp.kind == "pna":
            # W_out cat(a, amp a, att a) = W_0 a + amp (W_1 a) + att (W_2 a): one [V, H 4m] x [H 4m, 3 hidden] product and a
            # per-node combination, instead of writing and re-reading the [V, H 12m] (3072-wide) concatenation
            H, q, hid = self.mp.H, 4 * self.mp.msg, self.config.hidden_dim
            w = self.msg_out_projection.weight.view(hid, H, 3, q).permute(2, 0, 1, 3).reshape(3 * hid, H * q)
            p = dense.linear(self.mp(x, plan, scale=False), w)
            fused = (_FUSED_BLOCK and x.is_cuda and x.dtype == torch.float32 and self.config.use_rezero_scaling and self.boom_layer is not None
                     and self.config.dropout_rate == 0.0 and hid % 64 == 0 and hid <= 256 and isinstance(self.boom_norm_layer, nn.LayerNorm))
            if fused:
                # GPU: the combination, the ReZero residual and the BOOM layer norm in one kernel (csrc/block.h; no fallback on the
                # GPU for this configuration: a missing library raises)
                ln = self.boom_norm_layer
                x1, h = _BlockCombine.apply(p, x, plan.amplify.reshape(-1), plan.attenuate.reshape(-1), self.msg_out_projection.bias,
                                            self.alpha, ln.weight, ln.bias, ln.eps)
                return torch.addcmul(x1, self.alpha, self.boom_layer(h))          # x1 + alpha BOOM(LN(x1))
            new = p[:, :hid] + plan.amplify.to(x.dtype) * p[:, hid:2 * hid] + plan.attenuate.to(x.dtype) * p[:, 2 * hid:] \
                + self.msg_out_projection.bias
            new = self.dropout_layer(new)
        else:
            new = self.dropout_layer(self.msg_out_projection(self.mp(x, plan)))
        if self.config.use_rezero_scaling:
            new = self.alpha * new
        x = x + new
        if self.boom_layer is not None:
            boomed = self.dropout_layer(self.boom_layer(self.boom_norm_layer(x)))
            if self.config.use_rezero_scaling:
                boomed = self.alpha * boomed
            x = x + boomed
        return x


class GNN(nn.Module):
    def __init__(self, config: GNNConfig):
        super().__init__()
        self.config = config
        self.gnn_blocks = nn.ModuleList(GNNBlock(config) for _ in range(config.num_layers))

    def forward(self, node_features: torch.Tensor, adj_lists: List[torch.Tensor], plan: Optional[_GraphPlan] = None) -> List[torch.Tensor]:
        bidir, pna = self.config.make_edges_bidirectional, self.config.type.lower() == "pna"
        if plan is None or not plan.matches(node_features.shape[0], bidir, pna, node_features.device, node_features.dtype, adj_lists):
            plan = _GraphPlan(adj_lists, node_features.shape[0], bidir, pna, node_features.dtype)
        cur, states = node_features, [node_features]
        for blk in self.gnn_blocks:
            cur = blk(cur, plan)
            states.append(cur)
        if getattr(self, "state_grads", None) is not None:     # diagnostics (tools/diag_gnn_states.py): gradient arriving at every node state
            for k, st in enumerate(states):
                if st.requires_grad:
                    st.register_hook(lambda g, k=k: self.state_grads.__setitem__(k, g.detach().clone()))
        return states


def _segment_softmax(scores: torch.Tensor, index: torch.Tensor, num_segments: int) -> torch.Tensor:
    """torch_scatter.scatter_softmax(scores, index, dim=0) (graph_readout.py:238)."""
    idx = index.view(-1, 1).expand_as(scores)
    mx = scores.new_full((num_segments, scores.shape[1]), float("-inf")).scatter_reduce_(0, idx, scores, reduce="amax", include_self=True)
    ex = torch.exp(scores - mx[index])
    den = scores.new_zeros(num_segments, scores.shape[1]).index_add_(0, index, ex)
    return ex / den[index]


class _ReadoutPool(torch.autograd.Function):
    """Per-graph pooling of the combined read-out on the GPU (``adkf_readout_pool`` / ``_backward``, csrc/readout.h): segment
    softmax + weighted mean, sigmoid-weighted sum and max in ONE kernel, each per-graph sum in the fixed order of the graph's
    node list; the backward writes every element once.  Bit-reproducible, unlike ``index_add_`` and the backward of a gather."""

    @staticmethod
    def forward(ctx, s_mean, v_mean, s_sum, v_sum, emb, node_to_graph, num_graphs, nh, hd, perm=None, rowptr=None):
        import ctypes as C

        from . import _lib
        lib = _lib.load()
        dev = emb.device
        V, D, G = emb.shape[0], emb.shape[1], int(num_graphs)
        s_mean, v_mean, s_sum, v_sum, emb = (t.contiguous() for t in (s_mean, v_mean, s_sum, v_sum, emb))
        n2g = node_to_graph.contiguous()
        if perm is None or rowptr is None:      # (no plan on the batch: three tiny launches and a host synchronisation)
            counts = torch.bincount(n2g, minlength=G)
            perm = torch.argsort(n2g, stable=True)
            rowptr = torch.cat((counts.new_zeros(1), torch.cumsum(counts, 0)))
        f32 = dict(dtype=torch.float32, device=dev)
        w_mean, w_sum = torch.empty(V, nh, **f32), torch.empty(V, nh, **f32)
        g_mean, g_sum, g_max = torch.empty(G, nh * hd, **f32), torch.empty(G, nh * hd, **f32), torch.empty(G, D, **f32)
        argmax = torch.empty(G, D, dtype=torch.int32, device=dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(lib.adkf_readout_pool(ptr(s_mean), ptr(v_mean), ptr(s_sum), ptr(v_sum), ptr(emb), ptr(perm), ptr(rowptr), V, G, nh, hd, D,
                                         ptr(w_mean), ptr(w_sum), ptr(g_mean), ptr(g_sum), ptr(g_max), ptr(argmax), st), "adkf_readout_pool")
        ctx.save_for_backward(v_mean, v_sum, w_mean, w_sum, g_mean, argmax, n2g)
        ctx.dims = (V, G, nh, hd, D)
        return g_mean, g_sum, g_max

    @staticmethod
    def backward(ctx, dg_mean, dg_sum, dg_max):
        import ctypes as C

        from . import _lib
        lib = _lib.load()
        v_mean, v_sum, w_mean, w_sum, g_mean, argmax, n2g = ctx.saved_tensors
        V, G, nh, hd, D = ctx.dims
        dev = v_mean.device
        dg_mean, dg_sum, dg_max = dg_mean.contiguous(), dg_sum.contiguous(), dg_max.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        d_s_mean, d_s_sum = torch.empty(V, nh, **f32), torch.empty(V, nh, **f32)
        d_v_mean, d_v_sum, d_emb = torch.empty(V, nh * hd, **f32), torch.empty(V, nh * hd, **f32), torch.empty(V, D, **f32)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(lib.adkf_readout_pool_backward(ptr(v_mean), ptr(v_sum), ptr(w_mean), ptr(w_sum), ptr(g_mean), ptr(argmax), ptr(n2g),
                                                  ptr(dg_mean), ptr(dg_sum), ptr(dg_max), V, G, nh, hd, D, ptr(d_s_mean), ptr(d_v_mean),
                                                  ptr(d_s_sum), ptr(d_v_sum), ptr(d_emb), st), "adkf_readout_pool_backward")
        return d_s_mean, d_v_mean, d_s_sum, d_v_sum, d_emb, None, None, None, None, None, None


class _ReadoutPoolHidden(torch.autograd.Function):
    """The same pooling taken BEFORE the last layer of the two value MLPs (``adkf_readout_pool_hidden`` / ``_backward``,
    csrc/readout.h): per graph and head the pooled hidden activations ``p[h, g, :] = sum_v w[v, h] r_v`` and the weight totals,
    so that the value layers multiply ``[G, K]`` matrices instead of ``[V, K]`` ones.  ``h_mean`` / ``h_sum`` may be column
    blocks of one activation tensor (row stride passed to the kernel: no copies).  Same order guarantees as ``_ReadoutPool``."""

    @staticmethod
    def forward(ctx, s_mean, h_mean, s_sum, h_sum, emb, num_graphs, nh, perm, rowptr):
        import ctypes as C

        from . import _lib
        lib = _lib.load()
        dev = emb.device
        V, D, G, K = emb.shape[0], emb.shape[1], int(num_graphs), h_mean.shape[1]
        s_mean, s_sum, emb = s_mean.contiguous(), s_sum.contiguous(), emb.contiguous()
        if h_mean.stride(1) != 1 or h_sum.stride(1) != 1 or h_mean.stride(0) != h_sum.stride(0):
            h_mean, h_sum = h_mean.contiguous(), h_sum.contiguous()
        ldh = h_mean.stride(0) if V > 1 else K
        f32 = dict(dtype=torch.float32, device=dev)
        w_mean, w_sum = torch.empty(V, nh, **f32), torch.empty(V, nh, **f32)
        p_mean, p_sum = torch.empty(nh, G, K, **f32), torch.empty(nh, G, K, **f32)
        wtot_mean, wtot_sum = torch.empty(G, nh, **f32), torch.empty(G, nh, **f32)
        g_max, argmax = torch.empty(G, D, **f32), torch.empty(G, D, dtype=torch.int32, device=dev)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(lib.adkf_readout_pool_hidden(ptr(s_mean), ptr(h_mean), ptr(s_sum), ptr(h_sum), ldh, ptr(emb), ptr(perm), ptr(rowptr), V, G, nh,
                                                K, D, ptr(w_mean), ptr(w_sum), ptr(p_mean), ptr(p_sum), ptr(wtot_mean), ptr(wtot_sum), ptr(g_max),
                                                ptr(argmax), st), "adkf_readout_pool_hidden")
        ctx.save_for_backward(h_mean, h_sum, w_mean, w_sum, argmax, perm, rowptr)
        ctx.dims = (V, G, nh, K, D, ldh)
        ctx.mark_non_differentiable(wtot_mean)
        return p_mean, p_sum, wtot_mean, wtot_sum, g_max

    @staticmethod
    def backward(ctx, dp_mean, dp_sum, _dwtot_mean, dwtot_sum, dg_max):
        import ctypes as C

        from . import _lib
        lib = _lib.load()
        h_mean, h_sum, w_mean, w_sum, argmax, perm, rowptr = ctx.saved_tensors
        V, G, nh, K, D, ldh = ctx.dims
        dev = w_mean.device
        dp_mean, dp_sum, dwtot_sum, dg_max = dp_mean.contiguous(), dp_sum.contiguous(), dwtot_sum.contiguous(), dg_max.contiguous()
        f32 = dict(dtype=torch.float32, device=dev)
        d_s_mean, d_s_sum = torch.empty(V, nh, **f32), torch.empty(V, nh, **f32)
        d_h_mean, d_h_sum, d_emb = torch.empty(V, K, **f32), torch.empty(V, K, **f32), torch.empty(V, D, **f32)
        st = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        ptr = lambda t: C.c_void_p(t.data_ptr())
        _lib.check(lib.adkf_readout_pool_hidden_backward(ptr(h_mean), ptr(h_sum), ldh, ptr(w_mean), ptr(w_sum), ptr(argmax), ptr(perm), ptr(rowptr),
                                                         ptr(dp_mean), ptr(dp_sum), ptr(dwtot_sum), ptr(dg_max), V, G, nh, K, D, ptr(d_s_mean),
                                                         ptr(d_h_mean), ptr(d_s_sum), ptr(d_h_sum), ptr(d_emb), st),
                   "adkf_readout_pool_hidden_backward")
        return d_s_mean, d_h_mean, d_s_sum, d_h_sum, d_emb, None, None, None, None


class CombinedGraphReadout(nn.Module):
    """weighted-mean + weighted-sum (multi-head) + max pooling, then Linear(ReLU(cat))  (graph_readout.py:119-296).
    The two scoring MLPs and two value MLPs share their input, so their first layers run as one GEMM."""

    def __init__(self, node_dim: int, out_dim: int, num_heads: int, head_dim: int):
        super().__init__()
        self.nh, self.hd = num_heads, head_dim
        hid = num_heads * head_dim
        self.first = nn.Linear(node_dim, 4 * hid)            # [mean.score | mean.value | sum.score | sum.value] hidden layers
        self.mean_score_out = nn.Linear(hid, num_heads)
        self.mean_value_out = nn.Linear(hid, hid)
        self.sum_score_out = nn.Linear(hid, num_heads)
        self.sum_value_out = nn.Linear(hid, hid)
        self.mean_combination = nn.Linear(hid, out_dim, bias=False)
        self.sum_combination = nn.Linear(hid, out_dim, bias=False)
        self.max_combination = nn.Linear(node_dim, out_dim, bias=False)
        self.combination_layer = nn.Linear(3 * out_dim, out_dim, bias=False)

    def _project_pooled(self, p: torch.Tensor, wtot: torch.Tensor, layer: nn.Linear) -> torch.Tensor:
        """``g[g, h, :] = W2[h] p[h, g, :] + b2[h] wtot[g, h]`` for a value layer ``layer`` = (W2 [nh hd, hid], b2)."""
        G = p.shape[1]
        w = layer.weight.view(self.nh, self.hd, -1)                                    # [nh, hd, hid]
        g = torch.bmm(p, w.transpose(1, 2)).permute(1, 0, 2)                          # [G, nh, hd]
        g = g + wtot.unsqueeze(-1) * layer.bias.view(1, self.nh, self.hd)
        return g.reshape(G, self.nh * self.hd)

    def forward(self, node_embeddings: torch.Tensor, node_to_graph_id: torch.Tensor, num_graphs: int, plan: Optional[_GraphPlan] = None) -> torch.Tensor:
        V, hid = node_embeddings.shape[0], self.nh * self.hd
        h = F.relu(dense.linear(node_embeddings, self.first.weight, self.first.bias))
        h_ms, h_mv, h_ss, h_sv = h.split(hid, dim=1)
        if node_embeddings.is_cuda and node_embeddings.dtype == torch.float32 and self.nh <= 64:
            # GPU: one fused, order-fixed pooling kernel (no fallback: a missing library raises); more than 64 heads
            # (READOUT_MAX_HEADS, csrc/readout.h) is not a shape the kernels take: PyTorch's scatter ops below
            ok = (plan is not None and plan.perm_graph is not None and plan.num_graphs == num_graphs
                  and plan.perm_graph.device == node_embeddings.device and plan.perm_graph.shape[0] == V)
            segs = (plan.perm_graph, plan.rowptr_graph) if ok else (None, None)
            if _POOL_HIDDEN and hid <= 1024 and node_embeddings.shape[1] <= 2048:   # (the kernels' register budgets, csrc/readout.h)
                # pooling is linear in the last layer of the value MLPs: pool their HIDDEN activations per head, then nh products of
                # [G, hid] x [hid, hd] instead of [V, hid] x [hid, nh hd] over all nodes (csrc/readout.h)
                if not ok:      # (no plan on the batch: three tiny launches and a host synchronisation)
                    counts = torch.bincount(node_to_graph_id, minlength=num_graphs)
                    segs = (torch.argsort(node_to_graph_id, stable=True), torch.cat((counts.new_zeros(1), torch.cumsum(counts, 0))))
                p_mean, p_sum, wt_mean, wt_sum, g_max = _ReadoutPoolHidden.apply(self.mean_score_out(h_ms), h_mv, self.sum_score_out(h_ss), h_sv,
                                                                                 node_embeddings, num_graphs, self.nh, *segs)
                g_mean = self._project_pooled(p_mean, wt_mean, self.mean_value_out)
                g_sum = self._project_pooled(p_sum, wt_sum, self.sum_value_out)
            else:
                g_mean, g_sum, g_max = _ReadoutPool.apply(self.mean_score_out(h_ms), self.mean_value_out(h_mv), self.sum_score_out(h_ss),
                                                          self.sum_value_out(h_sv), node_embeddings, node_to_graph_id, num_graphs,
                                                          self.nh, self.hd, *segs)
        else:
            w_mean = _segment_softmax(self.mean_score_out(h_ms), node_to_graph_id, num_graphs)      # [V, heads]
            w_sum = torch.sigmoid(self.sum_score_out(h_ss))
            v_mean = self.mean_value_out(h_mv).view(V, self.nh, self.hd)
            v_sum = self.sum_value_out(h_sv).view(V, self.nh, self.hd)
            zeros = node_embeddings.new_zeros(num_graphs, hid)
            g_mean = zeros.index_add(0, node_to_graph_id, (w_mean.unsqueeze(-1) * v_mean).reshape(V, hid))
            g_sum = zeros.index_add(0, node_to_graph_id, (w_sum.unsqueeze(-1) * v_sum).reshape(V, hid))
            idx = node_to_graph_id.view(-1, 1).expand_as(node_embeddings)
            g_max = node_embeddings.new_zeros(num_graphs, node_embeddings.shape[1]).scatter_reduce_(
                0, idx, node_embeddings, reduce="amax", include_self=False)
        raw = torch.cat((self.mean_combination(g_mean), self.sum_combination(g_sum), self.max_combination(g_max)), dim=1)
        return self.combination_layer(F.relu(raw))


class GraphFeatureExtractor(nn.Module):
    def __init__(self, config: GraphFeatureExtractorConfig):
        super().__init__()
        self.config = config
        g, r = config.gnn_config, config.readout_config
        self.init_node_proj = nn.Linear(config.initial_node_feature_dim, g.hidden_dim, bias=False)
        self.gnn = GNN(g)
        node_dim = (g.num_layers + 1) * g.hidden_dim if r.use_all_states else g.hidden_dim
        if not r.readout_type.startswith("combined"):
            raise ValueError("only the reference's default 'combined' read-out is built")
        self.readout = CombinedGraphReadout(node_dim, r.output_dim, r.num_heads, r.head_dim)
        if config.output_norm == "off":
            self.final_norm_layer: Optional[nn.Module] = None
        elif config.output_norm == "layer":
            self.final_norm_layer = nn.LayerNorm(r.output_dim)
        elif config.output_norm == "batch":
            self.final_norm_layer = nn.BatchNorm1d(r.output_dim)
        else:
            raise ValueError(config.output_norm)

    def forward(self, batch) -> torch.Tensor:
        """``batch`` = anything with node_features, adjacency_lists, node_to_graph, num_graphs (FSMolBatch layout)."""
        plan = getattr(batch, "plan", None)
        states = self.gnn(self.init_node_proj(batch.node_features), list(batch.adjacency_lists), plan)
        node_repr = torch.cat(states, dim=-1) if self.config.readout_config.use_all_states else states[-1]
        out = self.readout(node_repr, batch.node_to_graph, batch.num_graphs, plan)
        if self.final_norm_layer is not None:
            out = self.final_norm_layer(out)
        return out

    # ---- checkpoint compatibility ------------------------------------------------------------
    def load_reference_state_dict(self, ref: Dict[str, torch.Tensor], prefix: str = "graph_feature_extractor.") -> None:
        """Maps a state dict with the reference's parameter names (per-tower, per-edge-type ``nn.Linear``s; separate
        read-out MLPs) onto the fused parameters of this module."""
        g = self.config.gnn_config
        get = lambda k: ref[prefix + k]
        own = {"init_node_proj.weight": get("init_node_proj.weight")}
        for b in range(g.num_layers):
            p = f"gnn.gnn_blocks.{b}."
            if g.use_rezero_scaling:
                own[p + "alpha"] = get(p + "alpha")
            for et in range(g.num_edge_types):
                for l in range(g.message_function_depth):
                    k = et * g.message_function_depth + l
                    ws = [get(f"{p}mp_layers.{h}.message_fns.{et}._layers.{2 * l}.weight").t() for h in range(g.num_heads)]
                    bs = [get(f"{p}mp_layers.{h}.message_fns.{et}._layers.{2 * l}.bias") for h in range(g.num_heads)]
                    own[f"{p}mp.weights.{k}"] = torch.stack(ws)
                    own[f"{p}mp.biases.{k}"] = torch.stack(bs)
            for name in ("msg_out_projection.weight", "msg_out_projection.bias", "mp_norm_layer.weight", "mp_norm_layer.bias"):
                own[p + name] = get(p + name)
            if g.intermediate_dim > 0:
                for name in ("boom_layer.linear1.weight", "boom_layer.linear1.bias", "boom_layer.linear2.weight",
                             "boom_layer.linear2.bias", "boom_norm_layer.weight", "boom_norm_layer.bias"):
                    own[p + name] = get(p + name)
        r = "readout."
        firsts_w, firsts_b = [], []
        for pool, tag in (("_weighted_mean_pooler", "mean"), ("_weighted_sum_pooler", "sum")):
            for mlp, kind in (("_scoring_module", "score"), ("_transformation_mlp", "value")):
                firsts_w.append(get(f"{r}{pool}.{mlp}._layers.0.weight"))
                firsts_b.append(get(f"{r}{pool}.{mlp}._layers.0.bias"))
                own[f"{r}{tag}_{kind}_out.weight"] = get(f"{r}{pool}.{mlp}._layers.2.weight")
                own[f"{r}{tag}_{kind}_out.bias"] = get(f"{r}{pool}.{mlp}._layers.2.bias")
            own[f"{r}{tag}_combination.weight"] = get(f"{r}{pool}._combination_layer.weight")
        own[r + "first.weight"] = torch.cat(firsts_w)
        own[r + "first.bias"] = torch.cat(firsts_b)
        own[r + "max_combination.weight"] = get(r + "_max_pooler._combination_layer.weight")
        own[r + "combination_layer.weight"] = get(r + "_combination_layer.weight")
        # output_norm = "batch": the running statistics are part of the checkpoint and decide what eval-mode features look
        # like - they are copied too, and a batch-norm checkpoint without them is refused (not silently evaluated with 0 / 1)
        for k in ("final_norm_layer.weight", "final_norm_layer.bias", "final_norm_layer.running_mean",
                  "final_norm_layer.running_var", "final_norm_layer.num_batches_tracked"):
            if prefix + k in ref:
                own[k] = get(k)
        missing, unexpected = self.load_state_dict(own, strict=False)
        unexpected = [k for k in unexpected]
        missing = [k for k in missing if "num_batches_tracked" not in k]
        if missing or unexpected:
            raise KeyError(f"reference checkpoint does not match: missing {missing}, unexpected {unexpected}")

    def reference_state_dict(self, prefix: str = "graph_feature_extractor.") -> Dict[str, torch.Tensor]:
        """The inverse of ``load_reference_state_dict``: this module's parameters under the reference's names and
        shapes (what ``ADKTModelTrainer.save_model`` would have written for the same weights)."""
        g = self.config.gnn_config
        own = {k: v.detach() for k, v in self.state_dict().items()}
        out: Dict[str, torch.Tensor] = {"init_node_proj.weight": own["init_node_proj.weight"]}
        for b in range(g.num_layers):
            p = f"gnn.gnn_blocks.{b}."
            if g.use_rezero_scaling:
                out[p + "alpha"] = own[p + "alpha"]
            for et in range(g.num_edge_types):
                for l in range(g.message_function_depth):
                    k = et * g.message_function_depth + l
                    for h in range(g.num_heads):
                        out[f"{p}mp_layers.{h}.message_fns.{et}._layers.{2 * l}.weight"] = own[f"{p}mp.weights.{k}"][h].t().contiguous()
                        out[f"{p}mp_layers.{h}.message_fns.{et}._layers.{2 * l}.bias"] = own[f"{p}mp.biases.{k}"][h].clone()
            names = ["msg_out_projection.weight", "msg_out_projection.bias", "mp_norm_layer.weight", "mp_norm_layer.bias"]
            if g.intermediate_dim > 0:
                names += ["boom_layer.linear1.weight", "boom_layer.linear1.bias", "boom_layer.linear2.weight",
                          "boom_layer.linear2.bias", "boom_norm_layer.weight", "boom_norm_layer.bias"]
            for name in names:
                out[p + name] = own[p + name]
        r = "readout."
        fw, fb = own[r + "first.weight"], own[r + "first.bias"]
        hid = fw.shape[0] // 4
        q = 0
        for pool, tag in (("_weighted_mean_pooler", "mean"), ("_weighted_sum_pooler", "sum")):
            for mlp, kind in (("_scoring_module", "score"), ("_transformation_mlp", "value")):
                out[f"{r}{pool}.{mlp}._layers.0.weight"] = fw[q * hid:(q + 1) * hid].clone()
                out[f"{r}{pool}.{mlp}._layers.0.bias"] = fb[q * hid:(q + 1) * hid].clone()
                out[f"{r}{pool}.{mlp}._layers.2.weight"] = own[f"{r}{tag}_{kind}_out.weight"]
                out[f"{r}{pool}.{mlp}._layers.2.bias"] = own[f"{r}{tag}_{kind}_out.bias"]
                q += 1
            out[f"{r}{pool}._combination_layer.weight"] = own[f"{r}{tag}_combination.weight"]
        out[r + "_max_pooler._combination_layer.weight"] = own[r + "max_combination.weight"]
        out[r + "_combination_layer.weight"] = own[r + "combination_layer.weight"]
        for k in ("final_norm_layer.weight", "final_norm_layer.bias", "final_norm_layer.running_mean",
                  "final_norm_layer.running_var", "final_norm_layer.num_batches_tracked"):
            if k in own:
                out[k] = own[k]
        return {prefix + k: v for k, v in out.items()}
