#!/usr/bin/env python
"""Config 3 of BASELINE.json, reported separately from bench.py: the full ADKF inner loop on one GPU with
the default deep-kernel model - GNN (PyTorch-ROCm, 10 PNA layers) + ECFP -> fc(2560 -> 2048 -> 2048) -> HIP GP fit + IFT
hypergradient - on synthetic FS-Mol-style tasks (random molecular graphs, 16-shot).  One extractor forward/backward per
meta-batch.  Usage: python tools/bench_c3.py [--tasks 16] [--support 16] [--query 128] [--steps 5]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from adkf_ift_amd.meta_batch import DKTBatch, MoleculeFeatures, collate_meta_batch, model_meta_step
from adkf_ift_amd.models import ADKTModel, ADKTModelConfig
from adkf_ift_amd.trainer import MetaStepConfig


from adkf_ift_amd.synthetic import random_molecules  # noqa: E402,F401  (kept importable from here: tools/determinism_probe.py)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tasks", type=int, default=16)     # tasks_per_batch of the reference
    ap.add_argument("--support", type=int, default=16)
    ap.add_argument("--query", type=int, default=128)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--gemm-tuning", choices=("off", "shipped", "tune"), default="shipped",
                    help="library-GEMM algorithm selection (adkf_ift_amd/gemm_tuning.py): off = hipBLASLt heuristic, shipped = the recorded "
                         "choices for these shapes, tune = measure now into --gemm-file")
    ap.add_argument("--gemm-file", default=None)
    ap.add_argument("--adam", choices=("fused", "foreach"), default="fused",
                    help="torch.optim.Adam implementation: fused = one multi-tensor kernel per parameter list (same update as the reference's "
                         "default; 32 -> 8 launches per step on the 85 tensors of the default model), foreach = PyTorch's default")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    if a.gemm_tuning != "off":
        from adkf_ift_amd.gemm_tuning import use_tuned_gemms
        use_tuned_gemms(a.gemm_file, tune=a.gemm_tuning == "tune")
    gen = torch.Generator().manual_seed(0)
    tasks = []
    for _ in range(a.tasks):
        s, q = random_molecules(a.support, gen), random_molecules(a.query, gen)
        tasks.append(DKTBatch(s, torch.rand(a.support, generator=gen) > 0.5, torch.randn(a.support, generator=gen),
                              q, torch.rand(a.query, generator=gen) > 0.5, torch.randn(a.query, generator=gen)))
    mb = collate_meta_batch(tasks).to(dev)
    torch.manual_seed(0)                            # the same random-init weights in every run (the step is bit-reproducible given them)
    model = ADKTModel(ADKTModelConfig()).to(dev)   # reference defaults: gnn+ecfp+fc, Matern-5/2, 2048-d features
    opt = torch.optim.Adam(model.feature_extractor_params(), lr=1e-4, **({"fused": True} if a.adam == "fused" else {}))
    cfg = MetaStepConfig(gp_kernel="matern", clip_value=1.0, inner_max_evals=200)
    for _ in range(a.warmup):
        model_meta_step(model, opt, mb, cfg)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        losses, _ = model_meta_step(model, opt, mb, cfg)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.steps
    print(json.dumps({"workload": f"C3: {a.tasks} tasks/step, {a.support}-shot, {a.query} query molecules, default GNN+ECFP+fc model "
                                  f"({sum(p.numel() for p in model.parameters()) / 1e6:.1f} M params), inner fit to convergence",
                      "tasks_per_s": a.tasks / dt, "ms_per_step": dt * 1e3, "nodes": int(mb.molecules.node_features.shape[0]),
                      "mean_loss": float(losses.mean()), "gemm_tuning": a.gemm_tuning, "adam": a.adam}))


if __name__ == "__main__":
    main()
