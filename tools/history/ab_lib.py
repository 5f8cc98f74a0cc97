"""Diagnostics: run the same seeded C2-shaped meta-batch through two builds of libadkf_gp.so (ADKF_LIB) and compare every
output bit by bit.   python tools/ab_lib.py <lib_a.so> <lib_b.so>      (each build runs in its own process)"""
import os, subprocess, sys
import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def worker(out):
    sys.path.insert(0, ROOT)
    import torch
    from adkf_ift_amd import gp_ops
    from adkf_ift_amd.synthetic import make_tasks
    dev = torch.device("cuda:0")
    tasks = make_tasks(16, 128, 256)
    Zs, Zq = tasks.features()
    phi, pri, l0 = gp_ops.init_params(Zs.to(dev))
    b = gp_ops.GPBatch(Zs.to(dev), tasks.y_s.to(dev), pri, "rbf", Z_q=Zq.to(dev), y_q=tasks.y_q.to(dev))
    phi, f, gn, ne, info = gp_ops.fit(b, phi, max_evals=20, exact_evals=True)
    o = gp_ops.ift_hypergrad(b, phi)
    res = {"l0": l0, "phi": phi, "f": f}
    for k, v in o.items():
        if torch.is_tensor(v):
            res[k] = v
    np.savez(out, **{k: v.detach().cpu().numpy() for k, v in res.items()})


if __name__ == "__main__":
    if sys.argv[1] == "--worker":
        worker(sys.argv[2])
        sys.exit(0)
    outs = []
    for i, lib in enumerate(sys.argv[1:3]):
        out = os.path.join(ROOT, "gpurun_out", f"ab_{i}.npz")
        subprocess.check_call([sys.executable, __file__, "--worker", out], env=dict(os.environ, ADKF_LIB=os.path.abspath(lib)))
        outs.append(np.load(out))
    a, b = outs
    for k in a.files:
        x, y = a[k], b[k]
        same = np.array_equal(x, y, equal_nan=True) if x.dtype.kind == "f" else np.array_equal(x, y)
        d = np.abs(x.astype(np.float64) - y.astype(np.float64)).max() if x.size else 0.0
        print(f"{k:12s} bit-identical={same}  max|a-b|={d:.3e}  max|a|={np.abs(x).max() if x.size else 0:.3e}")
