"""``functional_call`` with the signature of the reference's vendored copy
(fs_mol/utils/_stateless.py:79-119): run ``module(*args, **kwargs)`` with its parameters/buffers replaced by
the given tensors.  The reference vendored torch 1.11's implementation because it pinned torch 1.10; on
PyTorch-ROCm 2.x the same operation is ``torch.func.functional_call``, which this defers to."""
from __future__ import annotations

from typing import Any, Dict, Optional, Tuple

import torch
from torch.func import functional_call as _functional_call


def functional_call(module: torch.nn.Module, parameters_and_buffers: Dict[str, torch.Tensor], args,
                    kwargs: Optional[Dict[str, Any]] = None):
    if not isinstance(args, tuple):
        args = (args,)
    return _functional_call(module, parameters_and_buffers, args, kwargs or {})
