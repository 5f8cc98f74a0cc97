"""GEMM algorithm selection for the PyTorch-side layers of the deep-kernel model (GNN read-out, fc head, feature map).

hipBLASLt's default heuristic leaves time on the table for the tall-skinny float32 products of the molecular model (one
node-level read-out product alone is 0.49 TFLOP per step): with the algorithm measured per shape, the C3 step goes from
73.6 to 63.3 ms on one MI355X.  PyTorch ships the mechanism (``torch.cuda.tunable``, "TunableOp");
this module only wires it up:

* ``use_tuned_gemms()``          - apply recorded choices; shapes that are not in the file keep the default heuristic
* ``use_tuned_gemms(tune=True)`` - additionally measure every new shape once (tens of ms per shape) and append it to the
                                   file; meant for a warm-up pass over a dataset's batch shapes, not for steady state

The arithmetic is unchanged (float32 in, float32 accumulate): only which library kernel runs.  The file shipped in
``tuning/gfx950_gemm.csv`` holds the shapes of tools/bench_c3.py and of bench.py (C2) and is tied, by TunableOp's own validators, to this
image's hipBLASLt / rocBLAS build and to gfx950; on any other stack it is ignored and the default heuristic applies.
The hand-written HIP path (libadkf_gp) is not affected."""
from __future__ import annotations

import os
from typing import Optional

import torch

SHIPPED = os.path.join(os.path.dirname(os.path.abspath(__file__)), "tuning", "gfx950_gemm.csv")


def use_tuned_gemms(results_file: Optional[str] = None, tune: bool = False) -> str:
    import torch.cuda.tunable as tunable

    path = results_file or SHIPPED
    if tune and path == SHIPPED:
        raise ValueError("tuning appends to the results file: pass a writable results_file of your own, not the shipped one")
    if not tune and not os.path.exists(path):
        raise FileNotFoundError(path)
    if path == SHIPPED:   # TunableOp may rewrite its file when the process ends: let it have a private copy, never the tracked file
        import shutil, tempfile
        fd, private = tempfile.mkstemp(prefix="adkf_gemm_", suffix=".csv")
        os.close(fd)
        shutil.copyfile(path, private)
        path = private
        import atexit
        atexit.register(lambda f=private: os.path.exists(f) and os.remove(f))   # one private copy per process, removed with it
        try:
            tunable.write_file_on_exit(False)   # nothing is tuned in this mode: there is nothing to write back
        except Exception:
            pass
    tunable.enable(True)
    tunable.tuning_enable(bool(tune))
    tunable.set_filename(path, insert_device_ordinal=False)
    if os.path.exists(path):
        tunable.read_file(path)
    return path


def disable() -> None:
    import torch.cuda.tunable as tunable

    tunable.enable(False)
