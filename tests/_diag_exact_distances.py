"""Diagnostics (not collected): does the accuracy of the squared distances decide the float32 error of dL/dZ on the worst task of the
point-permutation batch?  The workspace's D^2 blocks are overwritten with float64-computed, float32-rounded ones (layout:
csrc/adkf_gp.hip::carve) before the fit and the hypergradient run on them (REUSE_DIST)."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run(dev, Zs, ys, Zq, yq, phi=None, exact=False):
    from adkf_ift_amd import gp_ops
    T, N, d = Zs.shape
    pri = torch.empty(T, 4, device=dev)
    b = gp_ops.GPBatch(Zs, ys, pri, "rbf", Z_q=Zq, y_q=yq)
    phi0, l0 = gp_ops.init_params_batch(b)
    if exact:
        ws, _ = b.workspace()
        flat = ws.view(torch.uint8)
        al = lambda n: (n * 4 + 255) // 256 * 256
        off = al(T * d)
        zs, zq = Zs.double(), Zq.double()
        for X, Y in ((zs, zs), (zq, zs), (zq, zq)):
            D = (X.unsqueeze(2) - Y.unsqueeze(1)).pow(2).sum(-1).float().contiguous()
            nb = D.numel() * 4
            flat[off:off + nb] = D.view(torch.uint8).flatten()
            off += al(D.numel())
    b.flags = gp_ops.REUSE_DIST
    if phi is None:
        phi, f, gn, ne, info = gp_ops.fit(b, phi0, max_evals=20, exact_evals=True)
    else:
        f, _, _, info = gp_ops.mll_value_grad(b, phi)
    b.flags = gp_ops.REUSE_DIST | gp_ops.REUSE_INNER
    out = gp_ops.ift_hypergrad(b, phi)
    return dict(phi=phi, dZ_s=out["dZ_s"], dZ_q=out["dZ_q"])


def main():
    from adkf_ift_amd.synthetic import make_tasks
    from oracle import gp_oracle as O
    dev = torch.device("cuda:0")
    T, N, d = 256, 128, 256
    tasks = make_tasks(T, N, d, first_task=900)
    Zs, Zq = (z.to(dev) for z in tasks.features())
    ys, yq = tasks.y_s.to(dev), tasks.y_q.to(dev)
    g = torch.Generator().manual_seed(2)
    ps, pq = torch.randperm(N, generator=g).to(dev), torch.randperm(N, generator=g).to(dev)
    for exact in (False, True):
        a = run(dev, Zs, ys, Zq, yq, exact=exact)
        b = run(dev, Zs[:, ps].contiguous(), ys[:, ps].contiguous(), Zq[:, pq].contiguous(), yq[:, pq].contiguous(), phi=a["phi"], exact=exact)
        for t in (199, 148, 19):
            p0, opri = O.init_phi(Zs[t].cpu().double(), False, True)
            q = O.full_reference_quantities(Zs[t].cpu(), ys[t].cpu(), Zq[t].cpu(), yq[t].cpu(), a["phi"][t].cpu().double(), opri, 0)
            for name, key, perm in (("dZ_s", "dZs_total", ps), ("dZ_q", "dZq_total", pq)):
                ref = q[key]; m = np.abs(ref).max()
                inv = torch.argsort(perm).cpu().numpy()
                ea = np.abs(a[name][t].cpu().numpy().astype(np.float64) - ref).max() / m
                eb = np.abs(b[name][t].cpu().numpy().astype(np.float64)[inv] - ref).max() / m
                print("exact D2" if exact else "device D2", "task", t, name, "original %.2e  permuted %.2e" % (ea, eb))


if __name__ == "__main__":
    main()
