"""Run the C2 benchmark loop and capture the first task whose GP section produces a non-finite number or a
non-zero info (how tests/golden/fit_noise_floor_task.npz was found; with the optimiser fixed it reports none)."""
import math, sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adkf_ift_amd import gp_ops
from adkf_ift_amd.synthetic import LinearFeatureMap, make_tasks
from adkf_ift_amd.trainer import ClipAdam, MetaStepConfig, meta_step, HipGPBackend
dev = torch.device("cuda:0")
tasks = make_tasks(256, 128, 256, N_q=128)
X_s, X_q, y_s, y_q = (a.to(dev) for a in (tasks.X_s, tasks.X_q, tasks.y_s, tasks.y_q))
W = tasks.W.to(dev).clone().requires_grad_(True)
opt = ClipAdam([W], lr=1e-4)
cfg = MetaStepConfig(gp_kernel="rbf", inner_max_evals=20, inner_exact_evals=True, clip_value=1.0)
f = LinearFeatureMap(X_s, X_q, W)

class Spy(HipGPBackend):
    def run(self, *a, **k):
        self.last = super().run(*a, **k)
        return self.last
spy = Spy()
os.makedirs("gpurun_out", exist_ok=True)
for k in range(140):
    Wprev = W.detach().clone()
    losses, phi = meta_step(f, [W], opt, y_s, y_q, cfg, backend=spy)
    phi, f_out, dZs, dZq, info_fit, info = spy.last
    bad_phi = ~torch.isfinite(phi).all(1)
    bad_f = ~torch.isfinite(f_out)
    bad_dz = ~torch.isfinite(dZs.flatten(1)).all(1) | ~torch.isfinite(dZq.flatten(1)).all(1)
    if bad_phi.any() or bad_f.any() or bad_dz.any() or (info_fit != 0).any() or (info != 0).any():
        print("step", k, "bad phi", bad_phi.nonzero().flatten().tolist(), "bad f_out", bad_f.nonzero().flatten().tolist(),
              "bad dZ", bad_dz.nonzero().flatten().tolist(), "info_fit", info_fit.nonzero().flatten().tolist(), info_fit[info_fit != 0].tolist(),
              "info", info.nonzero().flatten().tolist(), info[info != 0].tolist())
        t = int((bad_phi | bad_f | bad_dz).nonzero()[0])
        print("task", t, "phi", phi[t].tolist(), "f_out", float(f_out[t]))
        import numpy as np
        with torch.no_grad():
            Zs_t = (X_s[t] @ Wprev) / math.sqrt(256)
        np.savez("gpurun_out/fit_noise_floor_task.npz", Z_s=Zs_t.cpu().numpy(), y_s=y_s[t].cpu().numpy())
        print("wrote gpurun_out/fit_noise_floor_task.npz (move it to tests/golden/ and run make_golden.py fit_noise)")
        break
else:
    print("no non-finite value in 140 steps")
