"""Diagnostics: the inner fit of the golden fixtures through the build named by ADKF_LIB: f, |grad|, evaluations."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from adkf_ift_amd import gp_ops
from test_gpu_parity import _batch
dev = torch.device("cuda:0")
gd = os.path.join(ROOT, "tests", "golden")
for name in ("gp_N32_Nq32_d64_k0_r0_s0", "gp_N32_Nq32_d64_k1_r1_s0", "gp_N128_Nq128_d256_k0_r0_s0", "gp_N128_Nq128_d256_k1_r1_s0", "gp_N64_Nq128_d96_k0_r0_s4", "gp_N8_Nq8_d4_k0_r0_s0"):
    g = np.load(os.path.join(gd, name + ".npz"))
    b, _, n, m = _batch(g, dev)
    phi0 = torch.tensor(g["phi0"], dtype=torch.float32)[None].to(dev)
    phi, f, gn, ne, info = gp_ops.fit(b, phi0, max_evals=200)
    print(name, "f", f[0].item(), "oracle", float(g["f_in"]), "gn", gn[0].item(), "evals", ne[0].item(), "phi", phi[0].tolist(), "oracle phi", g["phi_star"].tolist() if "phi_star" in g.files else None)
