"""Diagnostics (not collected by pytest): the point-permutation property of tests/test_gpu_properties.py task by task - how far the
two float32 evaluations are apart, and how far EACH is from the float64 oracle at the same parameters, for the worst tasks.
`python tests/_diag_point_permutation.py` on the GPU box; ADKF_X3=0 selects the FP32 distance kernel.  Uses the oracle."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from test_gpu_properties import _run  # noqa: E402


def main():
    from adkf_ift_amd.synthetic import make_tasks
    from oracle import gp_oracle as O
    dev = torch.device("cuda:0")
    T, N, d, kernel = 256, 128, 256, "rbf"
    tasks = make_tasks(T, N, d, first_task=900)
    Zs, Zq = (z.to(dev) for z in tasks.features())
    ys, yq = tasks.y_s.to(dev), tasks.y_q.to(dev)
    g = torch.Generator().manual_seed(2)
    ps, pq = torch.randperm(N, generator=g).to(dev), torch.randperm(N, generator=g).to(dev)
    a = _run(dev, Zs, ys, Zq, yq, kernel, 20)
    b = _run(dev, Zs[:, ps].contiguous(), ys[:, ps].contiguous(), Zq[:, pq].contiguous(), yq[:, pq].contiguous(), kernel, 0, phi=a["phi"])
    for name, x, y in (("dZ_s", b["dZ_s"], a["dZ_s"][:, ps]), ("dZ_q", b["dZ_q"], a["dZ_q"][:, pq])):
        gmax = y.abs().max().item()
        per_task = (x - y).abs().amax(dim=(1, 2))
        own = y.abs().amax(dim=(1, 2))
        worst = torch.argsort(per_task, descending=True)[:3].tolist()
        print(name, "batch metric %.3e" % (per_task.max().item() / gmax), "global max |dZ| %.3e" % gmax)
        for t in worst:
            print("   task", t, "diff / global max %.2e" % (per_task[t].item() / gmax), "diff / own max %.2e" % (per_task[t].item() / own[t].item()),
                  "own max / global max %.2f" % (own[t].item() / gmax))
            p0, opri = O.init_phi(Zs[t].cpu().double(), False, True)
            q = O.full_reference_quantities(Zs[t].cpu(), ys[t].cpu(), Zq[t].cpu(), yq[t].cpu(), a["phi"][t].cpu().double(), opri, 0)
            ref = q["dZs_total"] if name == "dZ_s" else q["dZq_total"]
            dev_a = (a[name][t].cpu().numpy().astype(np.float64))
            inv = torch.argsort(ps if name == "dZ_s" else pq).cpu().numpy()
            dev_b = b[name][t].cpu().numpy().astype(np.float64)[inv]
            m = np.abs(ref).max()
            print("      against float64 (own max): original %.2e   permuted %.2e" % (np.abs(dev_a - ref).max() / m, np.abs(dev_b - ref).max() / m))


if __name__ == "__main__":
    main()
