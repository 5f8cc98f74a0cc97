"""Checkpoint compatibility with the reference.

``ADKTModelTrainer.save_model`` (fs_mol/utils/adaptive_dkt_utils.py:221-238) writes
``{"model_config": <pickled trainer config dataclass>, "model_state_dict": ..., ["optimizer_state_dict"], ["epoch"]}``;
``load_model_weights`` (:240-257) copies every saved tensor into the same-named entry of the model's state dict.

* ``load_reference_checkpoint`` reads such a file WITHOUT the reference package being importable: classes pickled from
  ``fs_mol.*`` are materialised as plain attribute bags, the configuration is converted field by field, and the weights go
  through ``GraphFeatureExtractor.load_reference_state_dict`` (per-tower Linear layers -> the fused parameters here).
* ``reference_state_dict`` / ``save_model`` write the reference's parameter names and shapes, with the configuration as a
  plain dict (a file the reference can consume with ``load_model_weights`` after building its model from its own config).
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
