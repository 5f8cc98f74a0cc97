#!/usr/bin/env python
"""The reference's only published wall-clock protocol (adaptive_dkt_walltime.py - meta-testing on all
157 FS-Mol test tasks, support size 64, every remaining molecule as query, 1 run; ~121 s on the authors' CPU box),
replayed on synthetic tasks of that shape: default deep-kernel model, random molecular graphs, query sizes drawn from a
seeded log-normal (median ~200, clipped to [32, 2000] - FS-Mol test tasks are of this order; the real data is not
available here).  The clock covers collation on the host, host->device copies, ONE extractor forward per chunk of
tasks, the batched inner fit, prediction, and the sklearn metrics, i.e. everything but reading files.
Usage: python tools/bench_meta_test.py [--tasks 157] [--support 64] [--chunk 16]"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from adkf_ift_amd import evaluate as E
from adkf_ift_amd.meta_batch import DKTBatch
from adkf_ift_amd.models import ADKTModel, ADKTModelConfig


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tasks", type=int, default=157)
    ap.add_argument("--support", type=int, default=64)
    ap.add_argument("--chunk", type=int, default=16, help="tasks per library call / extractor forward")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    from adkf_ift_amd.synthetic import meta_test_tasks
    t0 = time.perf_counter()
    tasks, sizes = meta_test_tasks(a.tasks, a.support)
    t_gen = time.perf_counter() - t0
    model = ADKTModel(ADKTModelConfig()).to(dev)
    E.evaluate_tasks(model, tasks[:4], tasks_per_call=4)   # warm-up (first-touch of the library and the allocator)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    res = E.evaluate_tasks(model, tasks, tasks_per_call=a.chunk)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    agg = E.avg_metrics_over_tasks({k: [v] for k, v in res.items()})
    print(json.dumps({"workload": f"meta-test protocol: {a.tasks} synthetic tasks, support {a.support}, query sizes "
                                  f"{int(sizes.min())}..{int(sizes.max())} (mean {sizes.mean():.0f}), default model, {a.chunk} tasks per call",
                      "walltime_s": dt, "tasks_per_s": a.tasks / dt, "query_molecules": int(sizes.sum()),
                      "reference_published_cpu_walltime_s": 121.1, "synthetic_generation_s": t_gen,
                      "mean_avg_precision": agg["avg_precision"][0]}))


if __name__ == "__main__":
    main()
