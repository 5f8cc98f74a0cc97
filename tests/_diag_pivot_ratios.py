"""Diagnostics (not collected): how many tasks of a synthetic batch the float64 path would take at the threshold ADKF_R64_THRESHOLD
(read once by the library), for the batch of the point-permutation property test and for bench.py's C2 batch."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from test_gpu_properties import _run  # noqa: E402


def main():
    from adkf_ift_amd import gp_ops
    from adkf_ift_amd.synthetic import make_tasks
    dev = torch.device("cuda:0")
    for first in (900, 0, 5000):
        tasks = make_tasks(256, 128, 256, first_task=first)
        Zs, Zq = (z.to(dev) for z in tasks.features())
        ys, yq = tasks.y_s.to(dev), tasks.y_q.to(dev)
        pri = torch.empty(256, 4, device=dev)
        b = gp_ops.GPBatch(Zs, ys, pri, "rbf", Z_q=Zq, y_q=yq)
        phi0, l0 = gp_ops.init_params_batch(b)
        b.flags = gp_ops.REUSE_DIST
        phi, f, gn, ne, info = gp_ops.fit(b, phi0, max_evals=20, exact_evals=True)
        b.flags = gp_ops.REUSE_DIST | gp_ops.REUSE_INNER
        out = gp_ops.ift_hypergrad(b, phi)
        fl = gp_ops.double_path_tasks(b)
        idx = torch.nonzero(fl).flatten().tolist()
        print("threshold", os.environ.get("ADKF_R64_THRESHOLD", "30 (default)"), "first_task", first, "flagged", len(idx), idx[:20])


if __name__ == "__main__":
    main()
