"""Diagnostics: distribution of the number of evaluations of the run-to-convergence inner fit over a C2 meta-batch."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from adkf_ift_amd import gp_ops
from adkf_ift_amd.synthetic import make_tasks
dev = torch.device("cuda:0")
tasks = make_tasks(256, 128, 256)
Zs, Zq = tasks.features()
phi0, pri, _ = gp_ops.init_params(Zs.to(dev))
b = gp_ops.GPBatch(Zs.to(dev), tasks.y_s.to(dev), pri, "rbf")
phi, f, gn, ne, info = gp_ops.fit(b, phi0, max_evals=200)
ne = ne.cpu().numpy(); gn = gn.cpu().numpy()
print("evals: mean %.1f  median %d  p90 %d  p99 %d  max %d" % (ne.mean(), np.median(ne), np.percentile(ne, 90), np.percentile(ne, 99), ne.max()))
print("histogram:", np.bincount(ne)[:60].tolist())
order = np.argsort(-ne)[:10]
print("slowest tasks: evals", ne[order].tolist(), "gnorm", ["%.1e" % g for g in gn[order]])
phi20, f20, gn20, _, _ = gp_ops.fit(b, phi0, max_evals=20, exact_evals=True)
print("f(converged) - f(20 evals): max %.2e  mean %.2e" % ((f20 - f).max().item(), (f20 - f).mean().item()))
