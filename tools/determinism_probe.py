#!/usr/bin/env python
"""Run-to-run bit comparison of the deep-kernel path on the GPU ("determinism check by re-run bit-compare").

Stage 1: the default-width extractor, forward + backward, twice on identical inputs -> features and every parameter gradient
         compared with torch.equal; a mismatch is reported per parameter (max |difference| / max |entry|).
Stage 2: ``model_meta_step`` (extractor -> GP fit -> IFT hypergradient -> backward -> SGD step) twice from identical initial
         weights -> losses, phi, gradients and updated parameters compared likewise.
Usage: python tools/determinism_probe.py [--tasks 4] [--support 16] [--query 48] [--small]"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch

from adkf_ift_amd.meta_batch import DKTBatch, collate_meta_batch, meta_features, model_meta_step
from adkf_ift_amd.models import ADKTModel, ADKTModelConfig
from adkf_ift_amd.trainer import MetaStepConfig
from bench_c3 import random_molecules


def diff(a, b):
    if torch.equal(a, b):
        return 0.0
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-300))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tasks", type=int, default=4)
    ap.add_argument("--support", type=int, default=16)
    ap.add_argument("--query", type=int, default=48)
    ap.add_argument("--alpha", type=float, default=0.3, help="ReZero gate of every block (1e-7 at init hides the message passing)")
    ap.add_argument("--runs", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(0)
    tasks = []
    for _ in range(a.tasks):
        s, q = random_molecules(a.support, gen), random_molecules(a.query, gen)
        tasks.append(DKTBatch(s, torch.rand(a.support, generator=gen) > 0.5, torch.randn(a.support, generator=gen),
                              q, torch.rand(a.query, generator=gen) > 0.5, torch.randn(a.query, generator=gen)))
    mb = collate_meta_batch(tasks).to(dev)

    def fresh():
        torch.manual_seed(7)
        m = ADKTModel(ADKTModelConfig()).to(dev)
        with torch.no_grad():
            for blk in m.graph_feature_extractor.gnn.gnn_blocks:
                blk.alpha.fill_(a.alpha)
        return m

    report = {"nodes": int(mb.molecules.node_features.shape[0]), "stage1": {}, "stage2": {}}
    # ---- stage 1 ----
    ref = None
    for r in range(a.runs):
        m = fresh()
        Zs, Zq = meta_features(m, mb)
        w_s = torch.randn(Zs.shape, generator=torch.Generator().manual_seed(1)).to(dev)
        w_q = torch.randn(Zq.shape, generator=torch.Generator().manual_seed(2)).to(dev)
        ((Zs * w_s).sum() + (Zq * w_q).sum()).backward()
        torch.cuda.synchronize()
        cur = {"Z_s": Zs.detach().clone(), "Z_q": Zq.detach().clone()}
        cur.update({"grad:" + n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
        if ref is None:
            ref = cur
            continue
        for k in ref:
            d = diff(cur[k], ref[k])
            if d > 0.0:
                report["stage1"][k] = max(report["stage1"].get(k, 0.0), d)
    # ---- stage 2 ----
    ref = None
    for r in range(a.runs):
        m = fresh()
        params = list(m.feature_extractor_params())
        opt = torch.optim.SGD(params, lr=0.1)
        losses, phi = model_meta_step(m, opt, mb, MetaStepConfig(gp_kernel="matern", clip_value=1.0), check=True)
        torch.cuda.synchronize()
        cur = {"losses": losses.detach().clone(), "phi": phi.detach().clone()}
        cur.update({"grad:" + n: p.grad.detach().clone() for n, p in m.named_parameters() if p.grad is not None})
        cur.update({"param:" + n: p.detach().clone() for n, p in m.named_parameters()})
        if ref is None:
            ref = cur
            continue
        for k in ref:
            d = diff(cur[k], ref[k])
            if d > 0.0:
                report["stage2"][k] = max(report["stage2"].get(k, 0.0), d)
    report["stage1_bitwise"] = not report["stage1"]
    report["stage2_bitwise"] = not report["stage2"]
    print(json.dumps(report, indent=1))
    return 0 if report["stage1_bitwise"] and report["stage2_bitwise"] else 1


if __name__ == "__main__":
    sys.exit(main())
