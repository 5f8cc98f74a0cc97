import glob, os, sys, numpy as np, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
from test_gpu_parity import _batch, rel
from adkf_ift_amd import gp_ops
dev = torch.device('cuda:0')
res = []
for f in sorted(glob.glob('/root/repo/tests/golden/gp_*.npz')):
    g = np.load(f)
    b, phi, n, m = _batch(g, dev)
    out = gp_ops.ift_hypergrad(b, phi)
    fin, gin, dZin, info = gp_ops.mll_value_grad(b, phi, want_dZ=True)
    H = out['H'][0].double().cpu().numpy()
    res.append((rel(out['dZ_s'][0, :n].cpu().numpy(), g['dZs_total']), rel(out['v'][0].cpu().numpy(), g['v']),
                rel(out['g_phi'][0].cpu().numpy(), g['g_out']), rel(fin[0].item(), g['f_in']), np.linalg.cond(g['H']), os.path.basename(f)))
for r in sorted(res, reverse=True)[:8]:
    print('dZs %.2e v %.2e g_out %.2e f_in %.2e condH %.1e %s' % r)
