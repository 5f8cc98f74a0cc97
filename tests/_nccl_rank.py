"""One rank of tests/test_zz_gpu_nccl.py (started as a FRESH process by the test: the pytest process has already touched the GPU
and must neither fork nor exec).  Shards the harness fixture's tasks over the ranks, runs trainer.meta_step at the fixture's
phi through libadkf_gp.so with the gradient all-reduce over RCCL (backend "nccl"), and writes what this rank ended up with.

    python tests/_nccl_rank.py <fixture.npz> <out_prefix> <count_rank0,count_rank1,..> [stated_total]
env: RANK, LOCAL_RANK, WORLD_SIZE, MASTER_ADDR, MASTER_PORT
"""
import math
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


class FixedPhiBackend:
    """HipGPBackend without the inner fit (the same test double as tests/test_gpu_reference_pins.py)."""

    def __init__(self, phi):
        self.phi = phi

    def run(self, Z_s, y_s, Z_q, y_q, cfg, n_s=None, n_q=None, fit_events=None, out_dZ=None):
        from adkf_ift_amd import gp_ops
        priors = torch.empty(Z_s.shape[0], 4, dtype=torch.float32, device=Z_s.device)
        b = gp_ops.GPBatch(Z_s, y_s, priors, cfg.gp_kernel, Z_q=Z_q, y_q=y_q, n_s=n_s, n_q=n_q)
        gp_ops.init_params_batch(b, cfg.use_numeric_labels, cfg.use_lengthscale_prior)
        b.flags = gp_ops.REUSE_DIST
        out = gp_ops.ift_hypergrad(b, self.phi, ignore_grad_correction=cfg.ignore_grad_correction, out_dZ=out_dZ)
        return self.phi, out["f_out"], out["dZ_s"], out["dZ_q"], torch.zeros_like(out["info"]), out["info"]


def main():
    fixture, out_prefix, split = sys.argv[1], sys.argv[2], [int(x) for x in sys.argv[3].split(",")]
    stated_total = int(sys.argv[4]) if len(sys.argv) > 4 else None
    rank, world, local = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"]), int(os.environ["LOCAL_RANK"])
    backend = os.environ.get("ADKF_TEST_DIST_BACKEND", "nccl")   # "gloo": the one-card rehearsal (both ranks on cuda:0, reduction through the host)
    local = local % torch.cuda.device_count()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if backend == "nccl":
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    else:
        dist.init_process_group(backend, rank=rank, world_size=world)

    from adkf_ift_amd.synthetic import make_tasks
    from adkf_ift_amd.trainer import MetaStepConfig, meta_step

    g = np.load(fixture)
    N, d = int(g["N"]), int(g["d"])
    lo, cnt = sum(split[:rank]), split[rank]
    tasks = make_tasks(cnt, N, d, first_task=500 + lo).to(dev)
    W = tasks.W.clone().requires_grad_(True)
    opt = torch.optim.SGD([W], lr=0.5)
    feats = lambda: (tasks.X_s @ W / math.sqrt(d), tasks.X_q @ W / math.sqrt(d))
    W0 = W.detach().clone()
    uneven = len(set(split)) > 1
    cfg = MetaStepConfig(gp_kernel="rbf", clip_value=1.0, uneven_shards=uneven and stated_total is None, global_tasks=stated_total)
    phi = torch.tensor(g["phi"][lo:lo + cnt], dtype=torch.float32).to(dev)
    losses, _ = meta_step(feats, [W], opt, tasks.y_s, tasks.y_q, cfg, backend=FixedPhiBackend(phi), distributed=True, check=True)
    torch.cuda.synchronize()
    np.savez(f"{out_prefix}_rank{rank}.npz", grad=W.grad.cpu().numpy(), step=((W0 - W.detach()) / 0.5).cpu().numpy(),
             losses=losses.cpu().numpy(), backend=np.array(dist.get_backend()))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
