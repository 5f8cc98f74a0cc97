"""Diagnostics (not collected by pytest): replay ONE case of tests/test_gpu_stress.py and print the error of every output of every
task against the float64 oracle.  `python tests/_diag_stress_case.py CASE` on the GPU box; environment switches of the library
(ADKF_X3=0 ...) apply.  Test infrastructure: uses the oracle."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from test_gpu_stress import _random_case, _rel  # noqa: E402
from _stress_oracle import oracle_bundle  # noqa: E402


def main(target):
    from adkf_ift_amd import gp_ops
    from adkf_ift_amd.synthetic import make_tasks
    dev = torch.device("cuda:0")
    rng = np.random.default_rng(20260)
    for case in range(target + 1):
        N, Nq, d, kind, regression, n_s, n_q = _random_case(rng)
    print(dict(case=case, N=N, Nq=Nq, d=d, kind=kind, regression=regression, n_s=n_s, n_q=n_q))
    tasks = make_tasks(3, N, d, N_q=Nq, regression=regression, first_task=100 * case)
    Zs, Zq = tasks.features()
    Zs, Zq, ys, yq = Zs.clone(), Zq.clone(), tasks.y_s.clone(), tasks.y_q.clone()
    for t in range(3):
        Zs[t, n_s[t]:] = 7.5; ys[t, n_s[t]:] = -3.0
        Zq[t, n_q[t]:] = -2.5; yq[t, n_q[t]:] = 9.0
    pri = torch.empty(3, 4, device=dev)
    b = gp_ops.GPBatch(Zs.to(dev), ys.to(dev), pri, kind, Z_q=Zq.to(dev), y_q=yq.to(dev),
                       n_s=torch.tensor(n_s, dtype=torch.int32), n_q=torch.tensor(n_q, dtype=torch.int32))
    phi0, l0 = gp_ops.init_params_batch(b, regression, True)
    b.flags = gp_ops.REUSE_DIST
    phi, f_in, gn, ne, info = gp_ops.fit(b, phi0, 200)
    b.flags = gp_ops.REUSE_DIST | gp_ops.REUSE_INNER
    out = gp_ops.ift_hypergrad(b, phi)
    flagged = gp_ops.double_path_tasks(b).tolist()
    mean, var, _, info = gp_ops.predict(b, phi)
    for t in range(3):
        n, m = n_s[t], n_q[t]
        got = {"f_in": f_in[t].item(), "H": out["H"][t].cpu().numpy(), "f_out": out["f_out"][t].item(),
               "g_out": out["g_phi"][t].cpu().numpy(), "v": out["v"][t].cpu().numpy(),
               "dZs_total": out["dZ_s"][t, :n].cpu().numpy(), "dZq_total": out["dZ_q"][t, :m].cpu().numpy(),
               "pred_mean": mean[t, :m].cpu().numpy(), "pred_var": var[t, :m].cpu().numpy()}
        o = oracle_bundle((Zs[t, :n].clone(), ys[t, :n].clone(), Zq[t, :m].clone(), yq[t, :m].clone(), phi[t].cpu().clone(), kind, regression))
        print("task", t, "n", n, "m", m, "cond %.1e" % o["cond"], "float64 path:", flagged[t], "phi", phi[t].tolist())
        for k, v in got.items():
            print("   %-10s err %.2e   float32 restatement %.2e   slack %.1f" % (k, _rel(v, o["q"][k]), o["e32"][k], o["slack"].get(k, 1.0)))


if __name__ == "__main__":
    main(int(sys.argv[1]))
