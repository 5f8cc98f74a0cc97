"""Checkpoint compatibility with the reference.

``ADKTModelTrainer.save_model`` (fs_mol/utils/adaptive_dkt_utils.py:221-238) writes
``{"model_config": <pickled trainer config dataclass>, "model_state_dict": ..., ["optimizer_state_dict"], ["epoch"]}``;
``load_model_weights`` (:240-257) copies every saved tensor into the same-named entry of the model's state dict.

* ``load_reference_checkpoint`` reads such a file WITHOUT the reference package being importable: classes pickled from
  ``fs_mol.*`` are materialised as plain attribute bags, the configuration is converted field by field, and the weights go
  through ``GraphFeatureExtractor.load_reference_state_dict`` (per-tower Linear layers -> the fused parameters here).
* ``reference_state_dict`` / ``save_model`` write the reference's parameter names and shapes, with the configuration as a
  plain dict (a file the reference can consume with ``load_model_weights`` after building its model from its own config).
* ``load_model_weights`` (:240-258) additionally restores the optimiser state - written by this package (fused parameters) or by
  the reference (one Adam slot per tower / edge-type ``nn.Linear``: mapped through the same name mapping as the weights).
* ``load_model_gnn_weights`` (:261-303): warm start of the GNN part only from a pre-trained extractor checkpoint, an optimiser
  with the loaded group at learning_rate / 10, and the 100-step linear warm-up of that group (``linear_warmup``,
  fs_mol/models/abstract_torch_fsmol_model.py:169-172).
"""
from __future__ import annotations

import pickle
import types
from dataclasses import asdict, fields, is_dataclass
from typing import Any, Dict, Optional

import torch

from .gnn import GNNConfig, GraphFeatureExtractorConfig, GraphReadoutConfig
from .models import ADKTModel, ADKTModelConfig

GP_PARAM_NAMES = ("gp_likelihood.noise_covar.raw_noise", "gp_model.covar_module.raw_outputscale",
                  "gp_model.covar_module.base_kernel.raw_lengthscale")


class _Bag:
    """Stand-in for a class that cannot be imported: keeps whatever state the pickle carries."""

    def __init__(self, *args, **kwargs):
        self.__dict__.update(kwargs)

    def __setstate__(self, state):
        if isinstance(state, tuple):  # (dict, slots)
            for part in state:
                if part:
                    self.__dict__.update(part)
        elif state:
            self.__dict__.update(state)


# Globals a checkpoint written by ``ADKTModelTrainer.save_model`` legitimately refers to.  Everything under ``fs_mol.``
# (the reference's config dataclasses and enums, not importable here) becomes an attribute bag; any OTHER global is refused:
# a pickle is a program, and this loader exists to read files that come from elsewhere.
_ALLOWED_GLOBALS = {
    ("collections", "OrderedDict"), ("builtins", "set"), ("builtins", "frozenset"), ("builtins", "dict"), ("builtins", "list"),
    ("builtins", "tuple"), ("builtins", "int"), ("builtins", "float"), ("builtins", "bool"), ("builtins", "str"),
    ("builtins", "complex"), ("builtins", "slice"), ("builtins", "range"), ("builtins", "bytearray"),
    ("torch._utils", "_rebuild_tensor_v2"), ("torch._utils", "_rebuild_tensor"), ("torch._utils", "_rebuild_parameter"),
    ("torch._utils", "_rebuild_parameter_with_state"), ("torch._utils", "_rebuild_qtensor"),
    ("torch._utils", "_rebuild_device_tensor_from_numpy"), ("torch._tensor", "_rebuild_from_type_v2"),
    ("torch.serialization", "_get_layout"), ("torch", "Size"), ("torch", "device"), ("torch", "dtype"),
    ("torch.nn.parameter", "Parameter"),
    ("numpy.core.multiarray", "_reconstruct"), ("numpy._core.multiarray", "_reconstruct"), ("numpy", "ndarray"), ("numpy", "dtype"),
    ("numpy.core.multiarray", "scalar"), ("numpy._core.multiarray", "scalar"),
}


class _TolerantUnpickler(pickle.Unpickler):
    def find_class(self, module, name):
        if (module, name) in _ALLOWED_GLOBALS:
            return super().find_class(module, name)
        if module == "torch" and (name.endswith("Storage") or name in ("float32", "float64", "float16", "bfloat16", "int64",
                                                                       "int32", "int16", "int8", "uint8", "bool")):
            return super().find_class(module, name)      # torch.FloatStorage & co., dtype singletons
        if module == "torch.storage" and name in ("TypedStorage", "UntypedStorage", "_load_from_bytes"):
            if name == "_load_from_bytes":
                raise pickle.UnpicklingError("refusing torch.storage._load_from_bytes (nested pickle) in a checkpoint")
            return super().find_class(module, name)
        if module.split(".")[0] in ("fs_mol", "MoleculeNet", "chem_lib", "dpu_utils") or module in ("pathlib", "enum"):
            return type(name, (_Bag,), {"__module__": module})     # the reference's own classes: kept as attribute bags
        raise pickle.UnpicklingError(f"checkpoint refers to the global {module}.{name}, which a reference checkpoint has no "
                                     f"reason to contain; refusing to import it")


_tolerant_pickle = types.SimpleNamespace(Unpickler=_TolerantUnpickler, load=lambda f, **kw: _TolerantUnpickler(f, **kw).load(),
                                         __name__="pickle")


def _get(obj, name, default):
    if obj is None:
        return default
    if isinstance(obj, dict):
        return obj.get(name, default)
    return getattr(obj, name, default)


def _sub_config(cls, src):
    return cls(**{f.name: _get(src, f.name, getattr(cls(), f.name)) for f in fields(cls)})


def convert_config(ref_config) -> ADKTModelConfig:
    """Reference ``ADKTModelConfig`` / ``ADKTModelTrainerConfig`` (object, attribute bag or dict) -> this package's config."""
    gfe = _get(ref_config, "graph_feature_extractor_config", None)
    d = ADKTModelConfig()
    gcfg = GraphFeatureExtractorConfig(
        initial_node_feature_dim=_get(gfe, "initial_node_feature_dim", GraphFeatureExtractorConfig().initial_node_feature_dim),
        gnn_config=_sub_config(GNNConfig, _get(gfe, "gnn_config", None)),
        readout_config=_sub_config(GraphReadoutConfig, _get(gfe, "readout_config", None)),
        output_norm=_get(gfe, "output_norm", "off"))
    return ADKTModelConfig(
        graph_feature_extractor_config=gcfg,
        used_features=_get(ref_config, "used_features", d.used_features),
        use_ard=_get(ref_config, "use_ard", d.use_ard),
        gp_kernel=_get(ref_config, "gp_kernel", d.gp_kernel),
        use_lengthscale_prior=_get(ref_config, "use_lengthscale_prior", d.use_lengthscale_prior),
        use_numeric_labels=_get(ref_config, "use_numeric_labels", d.use_numeric_labels),
        ignore_grad_correction=_get(ref_config, "ignore_grad_correction", d.ignore_grad_correction),
        fc_hidden_dim=_get(ref_config, "fc_hidden_dim", d.fc_hidden_dim), fc_out_dim=_get(ref_config, "fc_out_dim", d.fc_out_dim))


def load_reference_state_dict(model: ADKTModel, ref_sd: Dict[str, torch.Tensor]) -> None:
    """Reference-named tensors -> ``model``.  GP entries beyond the three raw parameters (the copies under ``mll.`` and
    ``gp_model.likelihood.``, prior / constraint buffers) are ignored: the GP tail is re-created for every task."""
    ref_sd = {k: (v.data if isinstance(v, torch.nn.Parameter) else v) for k, v in ref_sd.items()}
    if hasattr(model, "graph_feature_extractor"):
        model.graph_feature_extractor.load_reference_state_dict(ref_sd, prefix="graph_feature_extractor.")
    own = model.state_dict()
    with torch.no_grad():
        for name, value in ref_sd.items():
            if name.startswith("fc.") or name in GP_PARAM_NAMES:
                if name not in own:
                    raise KeyError(f"checkpoint entry {name} has no counterpart in the model")
                own[name].copy_(value.reshape(own[name].shape))
    expected_fc = [n for n in own if n.startswith("fc.")]
    missing = [n for n in expected_fc if n not in ref_sd]
    if missing:
        raise KeyError(f"checkpoint lacks {missing}")


def load_reference_checkpoint(path: str, device: Optional[torch.device] = None, config_overrides: Optional[Dict[str, Any]] = None):
    """``ADKTModelTrainer.build_from_model_file`` (adaptive_dkt_utils.py:303-326) for this package.
    Returns (model, checkpoint dict) - the dict still holds ``optimizer_state_dict`` / ``epoch`` when present."""
    ckpt = torch.load(path, map_location=device, pickle_module=_tolerant_pickle, weights_only=False)
    cfg = convert_config(ckpt["model_config"])
    if config_overrides:
        cfg = ADKTModelConfig(**{**{f.name: getattr(cfg, f.name) for f in fields(cfg)}, **config_overrides})
    model = ADKTModel(cfg)
    load_reference_state_dict(model, ckpt["model_state_dict"])
    if device is not None:
        model = model.to(device)
    return model, ckpt


def reference_state_dict(model: ADKTModel) -> Dict[str, torch.Tensor]:
    out: Dict[str, torch.Tensor] = {}
    if hasattr(model, "graph_feature_extractor"):
        out.update(model.graph_feature_extractor.reference_state_dict("graph_feature_extractor."))
    for name, value in model.state_dict().items():
        if name.startswith("fc.") or name in GP_PARAM_NAMES:
            out[name] = value.detach().clone()
    return out


def config_dict(cfg: ADKTModelConfig) -> Dict[str, Any]:
    d = {f.name: getattr(cfg, f.name) for f in fields(cfg)}
    g = d["graph_feature_extractor_config"]
    d["graph_feature_extractor_config"] = asdict(g) if is_dataclass(g) else g
    return d


def save_model(model: ADKTModel, path: str, optimizer: Optional[torch.optim.Optimizer] = None, epoch: Optional[int] = None) -> None:
    """Same file layout as the reference's ``save_model``; ``model_config`` is a plain dict (``convert_config`` reads it back)."""
    data: Dict[str, Any] = {"model_config": config_dict(model.config), "model_state_dict": reference_state_dict(model)}
    if optimizer is not None:
        data["optimizer_state_dict"] = optimizer.state_dict()
    if epoch is not None:
        data["epoch"] = epoch
    torch.save(data, path)


# ----------------------------------------------------------------------------------------------------------------------
# optimiser state (fs_mol/utils/adaptive_dkt_utils.py:255-258) and the GNN warm start (:261-303)
# ----------------------------------------------------------------------------------------------------------------------
def linear_warmup(cur_step: int, warmup_steps: int = 0) -> float:
    """fs_mol/models/abstract_torch_fsmol_model.py:169-172."""
    if cur_step >= warmup_steps:
        return 1.0
    return cur_step / warmup_steps


def _reference_param_names(ref_sd: Dict[str, torch.Tensor]):
    """Names of what the reference's ``feature_extractor_params()`` yields (fs_mol/models/adaptive_dkt.py:74-79: every
    parameter whose name does not start with ``gp_``), in registration order - the order of the Adam slots.  A state dict lists
    buffers as well: BatchNorm statistics are the only ones outside the GP tail."""
    skip = ("running_mean", "running_var", "num_batches_tracked")
    return [k for k in ref_sd if not k.startswith("gp_") and not k.startswith("mll.") and not k.endswith(skip)]


def restore_optimizer_state(model: ADKTModel, optimizer: torch.optim.Optimizer, opt_sd: Dict[str, Any],
                            ref_model_sd: Optional[Dict[str, torch.Tensor]] = None) -> str:
    """``load_model_weights``' second half (adaptive_dkt_utils.py:255-258).  Two layouts are understood:

    * "own": written by ``save_model`` of this package - one slot per (fused) parameter of ``feature_extractor_params()``:
      ``optimizer.load_state_dict``;
    * "reference": one slot per reference parameter (per-tower, per-edge-type Linear layers; separate read-out MLPs).  The
      ``exp_avg`` / ``exp_avg_sq`` tensors are arranged as state dicts under the reference's names (``ref_model_sd`` gives the
      slot order) and sent through the SAME mapping as the weights, which yields the moments of the fused parameters; ``step``
      and the hyper-parameters of the first group carry over.
    Returns which layout was found."""
    ours = [p for g in optimizer.param_groups for p in g["params"]]
    n_saved = sum(len(g["params"]) for g in opt_sd["param_groups"])
    state = opt_sd["state"]
    if n_saved == len(ours) and all(tuple(state[i]["exp_avg"].shape) == tuple(p.shape) for i, p in enumerate(ours) if i in state):
        optimizer.load_state_dict(opt_sd)
        return "own"
    if ref_model_sd is None:
        raise ValueError("optimizer state with the reference's parameter layout needs the checkpoint's model_state_dict for the slot order")
    names = _reference_param_names(ref_model_sd)
    if len(names) != n_saved:
        raise ValueError(f"optimizer state has {n_saved} slots, the checkpoint lists {len(names)} feature-extractor parameters")
    order = [i for g in opt_sd["param_groups"] for i in g["params"]]
    moments = {}
    for key in ("exp_avg", "exp_avg_sq"):
        as_sd = {}
        for slot, name in zip(order, names):
            ref_t = ref_model_sd[name]
            as_sd[name] = state[slot][key].reshape(ref_t.shape) if slot in state else torch.zeros_like(ref_t)
        scratch = ADKTModel(model.config)
        load_reference_state_dict(scratch, as_sd)
        moments[key] = dict(scratch.named_parameters())
    steps = [state[s]["step"] for s in order if s in state]
    step = steps[0] if steps else torch.tensor(0.0)
    name_of = {id(p): n for n, p in model.named_parameters()}
    for p in ours:
        n = name_of[id(p)]
        optimizer.state[p] = {"step": torch.as_tensor(float(step)), "exp_avg": moments["exp_avg"][n].detach().to(p).clone(),
                              "exp_avg_sq": moments["exp_avg_sq"][n].detach().to(p).clone()}
    # hyper-parameters group by group when the layouts agree; a saved state with ONE group next to a target with several (the lr / 10
    # group of load_model_gnn_weights) leaves the target's groups alone - overwriting them would undo the warm-start schedule and leave
    # ``initial_lr`` stale for its LambdaLR; several saved groups (a reference run saved after a GNN warm start: [other, gnn], built
    # from state_dict() order) do not map onto this package's registration order and are refused above by the slot count
    saved_groups = opt_sd["param_groups"]
    if len(saved_groups) == len(optimizer.param_groups):
        for g, gs in zip(optimizer.param_groups, saved_groups):
            for k in ("lr", "betas", "eps", "weight_decay", "amsgrad"):
                if k in gs and k in g:
                    g[k] = gs[k]
    return "reference"


def load_model_weights(model: ADKTModel, path: str, optimizer: Optional[torch.optim.Optimizer] = None,
                       device: Optional[torch.device] = None) -> Dict[str, Any]:
    """``ADKTModelTrainer.load_model_weights`` (adaptive_dkt_utils.py:240-258): weights into an EXISTING model, optimiser state
    into ``optimizer`` when the file carries one.  Returns the checkpoint dict."""
    ckpt = torch.load(path, map_location=device, pickle_module=_tolerant_pickle, weights_only=False)
    load_reference_state_dict(model, ckpt["model_state_dict"])
    if optimizer is not None and ckpt.get("optimizer_state_dict") is not None:
        restore_optimizer_state(model, optimizer, ckpt["optimizer_state_dict"], ckpt["model_state_dict"])
    return ckpt


def load_model_gnn_weights(model: ADKTModel, path: str, learning_rate: float, device: Optional[torch.device] = None):
    """``ADKTModelTrainer.load_model_gnn_weights`` (adaptive_dkt_utils.py:261-303): the graph feature extractor - everything
    but ``final_norm_layer`` - is loaded from a pre-trained extractor checkpoint whose ``model_state_dict`` names it WITHOUT the
    ``graph_feature_extractor.`` prefix (``readout_layer.`` of older files reads as ``readout.``); the rest of the model keeps
    its initialisation.  Returns ``(optimizer, lr_scheduler)``: Adam with two groups - everything else at ``learning_rate``, the
    loaded parameters at ``learning_rate / 10`` - and a ``LambdaLR`` that leaves the first group alone and ramps the loaded
    group up linearly over its first 100 steps.  Step the scheduler after every optimiser step (:412-413;
    ``model_meta_step(..., lr_scheduler=...)`` does)."""
    from functools import partial

    ckpt = torch.load(path, map_location=device, pickle_module=_tolerant_pickle, weights_only=False)
    pre = {k: (v.data if isinstance(v, torch.nn.Parameter) else v) for k, v in ckpt["model_state_dict"].items()}
    gfe = model.graph_feature_extractor
    ref = {}
    for name, value in pre.items():
        generic = "readout" + name[len("readout_layer"):] if name.startswith("readout_layer.") else name
        if "final_norm_layer" not in generic:
            ref["graph_feature_extractor." + generic] = value
    for k, v in gfe.state_dict().items():                 # not loaded (:272): the model's own output norm stays as it is
        if k.startswith("final_norm_layer."):
            ref["graph_feature_extractor." + k] = v.detach().clone()
    gfe.load_reference_state_dict(ref, prefix="graph_feature_extractor.")
    gnn_params = [p for n, p in gfe.named_parameters() if "final_norm_layer" not in n]
    loaded = {id(p) for p in gnn_params}
    other_params = [p for p in model.feature_extractor_params() if id(p) not in loaded]
    optimizer = torch.optim.Adam([{"params": other_params, "lr": learning_rate}, {"params": gnn_params, "lr": learning_rate / 10}])
    scheduler = torch.optim.lr_scheduler.LambdaLR(optimizer, lr_lambda=[partial(linear_warmup, warmup_steps=0),
                                                                        partial(linear_warmup, warmup_steps=100)])
    return optimizer, scheduler
