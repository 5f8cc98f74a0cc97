#!/usr/bin/env python
"""bench.py - meta-tasks/sec of the ADKF-IFT inner-loop hot path on MI355X (BASELINE.json metric).

A "step" = one outer (meta) step over one meta-batch of synthetic tasks per GPU (SURVEY 8d, config C2:
256 tasks, N_support = N_query = 128, d = 256): feature map Z = X W / sqrt(d) (theta = W stands in for the
GNN), per-task median-heuristic re-initialisation, inner fit with exactly I = 20 MLL value+gradient
evaluations, predictive NLL + 3x3 Hessian + IFT mixed term -> dL/dZ, one backward through the feature map,
(all-reduce of the outer gradient over ranks), clip, Adam step.  Inputs are resident in HBM before the clock
starts.  One process per GPU; for N > 1 launch with torch.distributed.run (RCCL).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--tasks", type=int, default=256, help="tasks per GPU per step (weak scaling)")
    ap.add_argument("--n-support", type=int, default=128)
    ap.add_argument("--n-query", type=int, default=128)
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--inner-evals", type=int, default=20)
    ap.add_argument("--kernel", default="rbf", choices=["rbf", "matern"])
    ap.add_argument("--converge", action="store_true", help="run the inner fit to convergence instead of a fixed I")
    ap.add_argument("--graph", action="store_true", help="replay the GP section of the step from a captured HIP graph")
    ap.add_argument("--ard", action="store_true", help="ARD kernel (h = 2 + d inner parameters): device L-BFGS + HVP/CG; not the headline config")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist

    import __graft_entry__ as ge
    from adkf_ift_amd import gp_ops, roofline
    from adkf_ift_amd.synthetic import LinearFeatureMap, make_tasks
    from adkf_ift_amd.trainer import ClipAdam, GraphedGPBackend, MetaStepConfig, meta_step

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: the GP path has no CPU fallback")
    # rehearsal of the N > 1 path on a one-GPU box: ADKF_BENCH_BACKEND=gloo lets several ranks share a card (RCCL
    # refuses duplicate devices); the driver's runs use the default, one rank per GPU over RCCL
    backend_name = os.environ.get("ADKF_BENCH_BACKEND", "nccl")
    dev_index = local_rank if backend_name == "nccl" else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    if distributed:
        if backend_name == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend_name)
    if rank == 0:
        ge.build()
    if distributed:
        dist.barrier()
    else:
        ge.build()

    T, N, Nq, d, I = args.tasks, args.n_support, args.n_query, args.d, args.inner_evals
    tasks = make_tasks(T, N, d, N_q=Nq, first_task=rank * T)
    X_s, X_q, y_s, y_q = (a.to(dev) for a in (tasks.X_s, tasks.X_q, tasks.y_s, tasks.y_q))
    W = tasks.W.to(dev).clone().requires_grad_(True)
    opt = ClipAdam([W], lr=1e-4)  # fs_mol/adaptive_dkt_train.py --lr default; mean + clip + Adam in the library (2 launches)
    cfg = MetaStepConfig(gp_kernel=args.kernel, inner_max_evals=(200 if args.converge else I),
                         inner_exact_evals=not args.converge, clip_value=1.0, use_ard=args.ard)
    inv_sqrt_d = 1.0 / math.sqrt(d)
    backend = GraphedGPBackend() if args.graph else None

    features = LinearFeatureMap(X_s, X_q, W)  # one GEMM for support+query rows; chunked-bmm backward

    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    for a, b_ in ev:  # create the underlying hipEvents
        a.record()
        b_.record()

    def sync():
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        meta_step(features, [W], opt, y_s, y_q, cfg, distributed=distributed, backend=backend)
    sync()
    t0 = time.perf_counter()
    for k in range(args.steps):
        losses, phi = meta_step(features, [W], opt, y_s, y_q, cfg, distributed=distributed, fit_events=ev[k], backend=backend)
    t_host = time.perf_counter() - t0   # when the host finished ENQUEUEING the K steps (== dt would mean host-bound)
    sync()
    dt = time.perf_counter() - t0
    if distributed:
        tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    fit_ms = sum(a.elapsed_time(b_) for a, b_ in ev) / len(ev)

    # ---- parity of the metric's second half ("logML rel-err") on a few tasks, outside the timed region ----
    parity = None
    if rank == 0 and not args.no_parity:
        from oracle import gp_oracle as O
        with torch.no_grad():
            Zall = features()
            Zs, Zq = Zall[0], Zall[1]
        phi0, pri, _ = gp_ops.init_params(Zs)
        b = gp_ops.GPBatch(Zs, y_s, pri, args.kernel, Z_q=Zq, y_q=y_q)
        phi_f, f_in, gn, nev, info = gp_ops.fit(b, phi0, cfg.inner_max_evals, exact_evals=cfg.inner_exact_evals)
        out = gp_ops.ift_hypergrad(b, phi_f)
        kind = gp_ops.kernel_id(args.kernel)
        e_in = e_out = e_dz = 0.0
        for t in range(min(2, T)):
            p = O.Priors(*pri[t].double().cpu().tolist())
            q = O.full_reference_quantities(Zs[t].cpu(), y_s[t].cpu(), Zq[t].cpu(), y_q[t].cpu(), phi_f[t].double().cpu(), p, kind)
            e_in = max(e_in, abs(f_in[t].item() - q["f_in"]) / abs(q["f_in"]))
            e_out = max(e_out, abs(out["f_out"][t].item() - q["f_out"]) / abs(q["f_out"]))
            ref = torch.as_tensor(q["dZs_total"])
            e_dz = max(e_dz, float((out["dZ_s"][t].double().cpu() - ref).abs().max() / ref.abs().max()))
        parity = {"logml_rel_err": e_in, "outer_nll_rel_err": e_out, "ift_dZ_rel_err": e_dz,
                  "fit_max_grad": float(gn.max().item()), "fit_mean_evals": float(nev.float().mean().item())}

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ref_cpu_path
        kind = gp_ops.kernel_id(args.kernel)
        rate, n_done, cores, nfev = ref_cpu_path.time_tasks(tasks, kind, budget_s=args.cpu_baseline_seconds)
        cpu_baseline = {"value": rate, "unit": "tasks/s", "cores": cores, "kind": "port",
                        "sample": f"first {n_done} tasks of the same workload, sequential, float32 torch restatement of the "
                                  f"reference algorithm (SciPy L-BFGS-B to convergence, mean {nfev:.0f} evals; dense "
                                  f"Hessian + nested-Jacobian hypergradient), torch threads = {cores}"}

    if rank == 0:
        fl = roofline.flops_per_task(N, Nq, d, I)
        total_tasks = T * world * args.steps
        value = total_tasks / dt
        fit_flops = fl["inner_fit"] * T            # algorithmic FLOPs of ONE launch of the dominant kernel
        achieved = fit_flops / (fit_ms * 1e-3) / 1e12
        # HBM bytes of that launch from the PMC pass committed under profiles/ (rocprofv3 --pmc FETCH_SIZE and
        # --pmc WRITE_SIZE in separate runs of this same command; KB -> bytes; the k_inner loads are dword-wide, for
        # which the guide's x2 FETCH_SIZE correction is uncalibrated, so the raw counters are reported)
        traffic = None
        pmc = os.path.join(ROOT, "profiles", "r01_final_k_inner_pmc.json")
        if os.path.exists(pmc) and (T, N, Nq, d, I) == (256, 128, 128, 256, 20) and args.kernel == "rbf":
            with open(pmc) as fh:
                pm = json.load(fh)
            traffic = (pm["FETCH_SIZE_KB_per_launch"] + pm["WRITE_SIZE_KB_per_launch"]) * 1024.0
        cfg_name = {(256, 128, 256): "C2", (64, 32, 64): "C1", (8, 1024, 512): "C5"}.get((T, N, d), "custom")
        if args.ard:
            cfg_name += " with the ARD kernel (roofline FLOP model below is the non-ARD one: indicative only)"
        fit_kernel = ("ARD inner fit (all launches between the two events)" if args.ard else "k_inner (in-kernel quasi-Newton fit: kernel build + register-resident sweep per evaluation)" if N <= 128 else
                      "blocked inner fit (all launches between the two events: k_lg_build, k_lg_diag, panel/update MFMA GEMMs, "
                      "k_lg_traces, k_lg_advance per evaluation)")
        line = {
            "metric": "meta-tasks/sec (N_support=128, d=256)", "value": value, "unit": "tasks/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg_name}: {T} tasks/GPU/step, N_support={N}, N_query={Nq}, d={d}, kernel={args.kernel}, "
                                   f"inner fit = {'to convergence' if args.converge else f'exactly {I} MLL value+grad evals'}, "
                                   "IFT hypergradient, theta = W[d,d] linear feature map, Adam + clip 1.0",
                       "tasks_per_gpu": T, "parallelism": f"task-sharded dp{world}"},
            "whole_path_tflops": value * fl["total"] / 1e12,
            "whole_path_frac_of_fp32_peak": value * fl["total"] / 1e12 / (roofline.PEAK_FP32_TFLOPS * world),
            "roofline": {"kernel": fit_kernel,
                         "bound": "mfma", "achieved": achieved, "peak": roofline.PEAK_FP32_TFLOPS, "unit": "TFLOP/s",
                         "frac": achieved / roofline.PEAK_FP32_TFLOPS, "traffic": traffic,
                         "flops_per_launch": fit_flops, "avg_launch_ms": fit_ms},
            "host_enqueue_ms_per_step": t_host / args.steps * 1e3,
            "cpu_baseline": cpu_baseline,
            "parity": parity,
        }
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
