#!/usr/bin/env python
"""bench.py - meta-tasks/sec of the ADKF-IFT inner-loop hot path on MI355X (BASELINE.json metric).

A "step" = one outer (meta) step over one meta-batch of synthetic tasks per GPU (SURVEY 8d, config C2:
256 tasks, N_support = N_query = 128, d = 256): feature map Z = X W / sqrt(d) (theta = W stands in for the
GNN), per-task median-heuristic re-initialisation, inner fit with exactly I = 20 MLL value+gradient
evaluations, predictive NLL + 3x3 Hessian + IFT mixed term -> dL/dZ, one backward through the feature map,
(all-reduce of the outer gradient over ranks), clip, Adam step.  Inputs are resident in HBM before the clock
starts.  One process per GPU.

Launching.  ``python bench.py --gpus N`` with N > 1 and no WORLD_SIZE in the environment starts the N rank processes
itself (plain child processes, one GPU each through LOCAL_RANK, rendezvous on 127.0.0.1) BEFORE anything touches the
GPU, relays rank 0's JSON line and exits with the worst child status; the parent never initialises HIP and never
re-execs.  Under ``python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N`` the ranks are already
there (WORLD_SIZE is set) and each process is one rank; ``--gpus`` must then agree with WORLD_SIZE.

Scaling modes.  Default: weak (``--tasks`` per GPU).  ``--global-tasks G``: strong (G tasks split over the ranks: the C4
shape is ``--global-tasks 512 --gpus 8`` = 64 tasks per rank).

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import math
import os
import socket
import subprocess
import tempfile
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--tasks", type=int, default=256, help="tasks per GPU per step (weak scaling)")
    ap.add_argument("--global-tasks", type=int, default=0, help="strong scaling: this many tasks per step in total, split over the ranks")
    ap.add_argument("--n-support", type=int, default=128)
    ap.add_argument("--n-query", type=int, default=128)
    ap.add_argument("--d", type=int, default=256)
    ap.add_argument("--inner-evals", type=int, default=20)
    ap.add_argument("--kernel", default="rbf", choices=["rbf", "matern"])
    ap.add_argument("--regression", action="store_true",
                    help="standardised real labels and the numeric-label initialisation (noise 0.01: fs_mol/utils/gp_utils.py:17, "
                         "fs_mol/models/adaptive_dkt.py:112-119) instead of +-1 classification labels; the line then also reports "
                         "which fraction of the tasks took the float64 path for ill-conditioned tasks")
    ap.add_argument("--converge", action="store_true", help="headline loop runs the inner fit to convergence instead of a fixed I")
    ap.add_argument("--converge-steps", type=int, default=5, help="steps of the second, run-to-convergence timed loop (0: skip)")
    ap.add_argument("--graph", action="store_true", help="replay the GP section of the step from a captured HIP graph")
    ap.add_argument("--ard", action="store_true", help="ARD kernel (h = 2 + d inner parameters): device L-BFGS + HVP/CG; not the headline config")
    ap.add_argument("--cpu-baseline-seconds", type=float, default=15.0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-meta-test", action="store_true",
                    help="skip the `meta_test` object (the reference's published wall-clock protocol on synthetic molecular tasks)")
    ap.add_argument("--meta-test-tasks", type=int, default=157)
    ap.add_argument("--side-configs", choices=("auto", "on", "off"), default="auto",
                    help="the other BASELINE.json configurations as side objects of the line, measured AFTER the headline loop like `converged` / "
                         "`meta_test`: c1 (64 x 32 x 64), t512 (512 tasks/GPU), c5 (8 x 1024 x 512), c3 (default GNN+ECFP+fc model, 16-shot) and c3 at "
                         "the reference's CLI shape (support 64 / query 256).  auto: only on the default C2 command, one GPU")
    ap.add_argument("--gemm-tuning", choices=("off", "shipped"), default="shipped",
                    help="algorithm choice of the two library GEMMs of the theta = W feature map (adkf_ift_amd/gemm_tuning.py): "
                         "hipBLASLt's heuristic, or the recorded choice for these shapes (same float32 arithmetic)")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / collective rehearsal without a GPU: ranks rendezvous (gloo), all-reduce a dummy gradient, "
                         "time barriers and print a line with value null (tests/test_bench_launcher.py)")
    return ap.parse_args(argv)


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def launch_ranks(n: int, argv, timeout_s: float = 3000.0) -> int:
    """Parent of a self-launched multi-GPU run.  Touches no GPU API (torch is not even imported here).  Polls ALL children:
    the first rank that exits non-zero (or the wall-clock limit) ends the run - the siblings, which would wait for it in the
    rendezvous or in an all-reduce for ever, are terminated (these exact child processes) and that status is returned."""
    port = os.environ.get("MASTER_PORT") or str(_free_port())
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # rank 0's stdout is the result line: it goes to a temporary file (a pipe nobody drains could fill up and block the rank)
        out0 = tempfile.TemporaryFile() if r == 0 else None
        procs.append((subprocess.Popen([sys.executable, os.path.abspath(__file__)] + list(argv), env=env,
                                       stdout=out0 if r == 0 else subprocess.DEVNULL), out0))
    deadline = time.monotonic() + timeout_s
    rc = 0
    while True:
        codes = [p.poll() for p, _ in procs]
        bad = [c for c in codes if c not in (None, 0)]
        if bad:
            rc = bad[0]
            break
        if all(c == 0 for c in codes):
            break
        if time.monotonic() > deadline:
            rc = 124
            sys.stderr.write(f"bench.py: ranks still running after {timeout_s:.0f} s\n")
            break
        time.sleep(0.05)
    for p, _ in procs:
        if p.poll() is None:
            p.terminate()
    for p, _ in procs:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
            p.wait()
    f0 = procs[0][1]
    f0.seek(0)
    sys.stdout.write(f0.read().decode())
    sys.stdout.flush()
    return rc


def profile_json(name: str):
    """The newest committed PMC summary of that name (profiles/r05_<name>, else r04_<name>) and its path, or (None, None)."""
    for rnd in ("r05", "r04"):
        f = os.path.join(ROOT, "profiles", f"{rnd}_{name}")
        if os.path.exists(f):
            with open(f) as fh:
                return json.load(fh), f"profiles/{rnd}_{name}"
    return None, None


def metric_name(N: int, d: int) -> str:
    return f"meta-tasks/sec (N_support={N}, d={d})"


def dry_run(args, rank: int, world: int):
    import torch
    import torch.distributed as dist

    if os.environ.get("ADKF_BENCH_DRYRUN_FAIL_RANK") == str(rank):   # (tests/test_bench_launcher.py: a rank that dies before the rendezvous)
        raise SystemExit(3)
    if world > 1:
        dist.init_process_group(os.environ.get("ADKF_BENCH_BACKEND", "gloo"))
    T = args.global_tasks // world if args.global_tasks else args.tasks
    g = torch.full((args.d, args.d), float(rank + 1))
    t0 = time.perf_counter()
    for _ in range(args.steps):
        if world > 1:
            dist.all_reduce(g)
            dist.barrier()
    dt = time.perf_counter() - t0
    if rank == 0:
        print(json.dumps({"metric": metric_name(args.n_support, args.d), "value": None, "unit": "tasks/s", "n_gpus": world,
                          "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / max(args.steps, 1) * 1e3,
                          "scaling": "strong" if args.global_tasks else "weak", "dry_run": True,
                          "config": {"tasks_per_gpu": T, "parallelism": f"task-sharded dp{world}"}}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    args = parse()
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        raise SystemExit(launch_ranks(args.gpus, sys.argv[1:]))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
                         f"(or drop WORLD_SIZE and let bench.py start the ranks)")
    if args.global_tasks and args.global_tasks % world:
        raise SystemExit(f"--global-tasks {args.global_tasks} does not divide over {world} ranks")
    if args.dry_run:
        return dry_run(args, rank, world)

    import torch
    import torch.distributed as dist

    import __graft_entry__ as ge
    from adkf_ift_amd import gp_ops, roofline
    from adkf_ift_amd.synthetic import LinearFeatureMap, make_tasks
    from adkf_ift_amd.trainer import ClipAdam, GraphedGPBackend, MetaStepConfig, meta_step

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
    if args.gemm_tuning == "shipped":
        try:
            from adkf_ift_amd.gemm_tuning import use_tuned_gemms
            use_tuned_gemms()
        except Exception as e:   # the recorded choices are an optimisation of two library calls: never a reason to fail the run
            print(f"[bench] library-GEMM choices not applied ({type(e).__name__}: {e}); hipBLASLt heuristic in use", file=sys.stderr)
            args.gemm_tuning = "off"

    T = args.global_tasks // world if args.global_tasks else args.tasks
    N, Nq, d, I = args.n_support, args.n_query, args.d, args.inner_evals

    def sync():
        torch.cuda.synchronize(dev)
        if distributed:
            dist.barrier()
            torch.cuda.synchronize(dev)

    class Workload:
        """Synthetic tasks of one shape resident in HBM + the meta-step over them."""

        def __init__(self, T_, N_, Nq_, d_, regression=False, first_task=0):
            self.T, self.N, self.Nq, self.d = T_, N_, Nq_, d_
            self.tasks = make_tasks(T_, N_, d_, N_q=Nq_, regression=regression, first_task=first_task)
            self.X_s, self.X_q, self.y_s, self.y_q = (a.to(dev) for a in (self.tasks.X_s, self.tasks.X_q, self.tasks.y_s, self.tasks.y_q))
            self.W = self.tasks.W.to(dev).clone().requires_grad_(True)
            self.opt = ClipAdam([self.W], lr=1e-4)  # fs_mol/adaptive_dkt_train.py --lr default; mean + clip + Adam in the library (1 launch)
            self.features = LinearFeatureMap(self.X_s, self.X_q, self.W)  # one GEMM for support+query rows; chunked-bmm backward
            self.features.planes_from(self.opt)  # the optimiser's launch also writes the bfloat16 planes of the new W (no split launch)

        def timed_loop(self, cfg_, steps, warmup, events=None, backend=None, dist_=False):
            for _ in range(warmup):
                meta_step(self.features, [self.W], self.opt, self.y_s, self.y_q, cfg_, distributed=dist_, backend=backend)
            sync()
            t0 = time.perf_counter()
            for k in range(steps):
                meta_step(self.features, [self.W], self.opt, self.y_s, self.y_q, cfg_, distributed=dist_, backend=backend,
                          fit_events=events[k] if events else None)
            t_host = time.perf_counter() - t0   # when the host finished ENQUEUEING the steps (== dt would mean host-bound)
            sync()
            dt = time.perf_counter() - t0
            if dist_:
                tmax = torch.tensor([dt], device=dev, dtype=torch.float64)
                dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
                dt = float(tmax.item())
            return dt, t_host

    def make_events(n):
        ev_ = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(n)]
        for a, b_ in ev_:  # create the underlying hipEvents
            a.record()
            b_.record()
        return ev_

    wl = Workload(T, N, Nq, d, regression=args.regression, first_task=rank * T)
    tasks, X_s, X_q, y_s, y_q, W, features = wl.tasks, wl.X_s, wl.X_q, wl.y_s, wl.y_q, wl.W, wl.features

    def step_cfg(converge: bool, kernel=None, evals=None) -> MetaStepConfig:
        return MetaStepConfig(gp_kernel=kernel or args.kernel, inner_max_evals=(200 if converge else (evals or I)),
                              inner_exact_evals=not converge, clip_value=1.0, use_ard=args.ard, use_numeric_labels=args.regression)

    cfg = step_cfg(args.converge)
    backend = GraphedGPBackend() if args.graph else None

    def timed_loop(cfg_, steps, warmup, events=None):
        return wl.timed_loop(cfg_, steps, warmup, events, backend=backend, dist_=distributed)

    # HIP events recorded by the library on the launch stream right around the inner-fit kernel (adkf_fit_options_t).
    # Under --graph the captured launch carries no events, so the roofline object is omitted there.
    ev = make_events(args.steps) if not args.graph else None
    dt, t_host = timed_loop(cfg, args.steps, args.warmup, ev)
    fit_ms = sum(a.elapsed_time(b_) for a, b_ in ev) / len(ev) if ev else None

    # ---- SURVEY 8d "also report run-to-convergence": a second, short timed loop with the fit run to its stopping rules ----
    converged = None
    if args.converge_steps > 0 and not args.converge and not args.ard:
        cdt, _ = timed_loop(step_cfg(True), args.converge_steps, 2)
        converged = {"tasks_per_s": T * world * args.converge_steps / cdt, "ms_per_step": cdt / args.converge_steps * 1e3,
                     "steps": args.converge_steps}

    # ---- parity of the metric's second half ("logML rel-err") on a few tasks, outside the timed region ----
    parity = None
    if rank == 0 and not args.no_parity and args.ard:
        # ARD: the conjugate-gradient rounds of H v = grad_phi f_out at the fitted point (value parity of the ARD path:
        # tests/test_gpu_ard.py against the autograd fixtures)
        with torch.no_grad():
            feats = features()
            Zs, Zq = feats[0], feats[1]
        pri = torch.empty(T, 4, device=dev)
        b = gp_ops.GPBatch(Zs, y_s, pri, args.kernel, Z_q=Zq, y_q=y_q, ard=True)
        phi0, _ = gp_ops.init_params_batch(b, args.regression, True)
        phi_f, f_in, gn, nev, info = gp_ops.fit(b, phi0, cfg.inner_max_evals, exact_evals=cfg.inner_exact_evals)
        out = gp_ops.ift_hypergrad(b, phi_f)
        it0 = out["cg_iters"].float()
        parity = {"fit_max_grad": float(gn.max().item()), "fit_mean_evals": float(nev.float().mean().item()),
                  "cg_rounds": {"mean": float(it0.mean().item()), "max": float(it0.max().item())}}
    elif rank == 0 and not args.no_parity:
        from oracle import gp_oracle as O
        with torch.no_grad():
            feats = features()
            Zs, Zq = feats[0], feats[1]
        phi0, pri, _ = gp_ops.init_params(Zs, use_numeric_labels=args.regression)
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
        parity["float64_path_fraction"] = float(gp_ops.double_path_tasks(b).float().mean().item())
        if converged is not None:
            _, _, gn_c, nev_c, _ = gp_ops.fit(b, phi0, 200, exact_evals=False)
            converged["mean_evals"] = float(nev_c.float().mean().item())
            converged["fit_max_grad"] = float(gn_c.max().item())

    cpu_baseline = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ref_cpu_path
        kind = gp_ops.kernel_id(args.kernel)
        rate, n_done, cores, nfev = ref_cpu_path.time_tasks(tasks, kind, budget_s=args.cpu_baseline_seconds)
        cpu_baseline = {"value": rate, "unit": "tasks/s", "cores": cores, "kind": "port",
                        "sample": f"first {n_done} tasks of the same workload, sequential, float32 torch restatement of the "
                                  f"reference algorithm (SciPy L-BFGS-B to convergence, mean {nfev:.0f} evals; dense "
                                  f"Hessian + nested-Jacobian hypergradient), torch threads = {cores}"}

    # the second CPU baseline: the C++ twin of the C ABI (OpenMP over tasks, float64 inside), same unit of work
    # as the GPU step (fresh parameters, exactly I evaluations, IFT hypergradient) on a prefix of the same workload
    cpu_twin = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not args.ard:
        try:
            from oracle import cpu_twin as TW
            rate, n_done, threads = TW.time_tasks(tasks, gp_ops.kernel_id(args.kernel), I, budget_s=args.cpu_baseline_seconds, regression=args.regression)
            cpu_twin = {"value": rate, "unit": "tasks/s", "cores": threads, "kind": "twin",
                        "sample": f"first {n_done} tasks of the same workload through libadkf_gp_cpu.so (C++ twin of the C ABI, "
                                  f"adkf_ift_amd/csrc/cpu/adkf_gp_cpu.cpp: OpenMP over tasks, float64 inside): adkf_init_params -> "
                                  f"adkf_fit (exactly {I} evaluations) -> adkf_ift_hypergrad, {threads} threads"}
        except Exception as e:   # a missing host compiler must not cost the GPU line
            print(f"[bench] CPU twin not timed ({type(e).__name__}: {e})", file=sys.stderr)

    # ---- the reference's only PUBLISHED wall-clock protocol, outside the timed region (like `converged`): meta-testing on 157
    # tasks at support size 64, every other molecule of the task as query (fs_mol/adaptive_dkt_walltime.py:100-115,
    # fs_mol/utils/test_utils.py:236-350; 121 s on the authors' CPU box, visualize_results/visualize_classification.ipynb).
    # Here: synthetic molecular tasks of that shape through evaluate.evaluate_tasks (default 25 M-parameter model, random
    # weights) - one extractor forward per chunk of tasks, batched inner fit, prediction, sklearn metrics.  Context, not a
    # like-for-like comparison: no file reading, no trained weights, synthetic graphs.
    meta_test = None
    if rank == 0 and world == 1 and not args.no_meta_test and not args.ard:
        try:
            from adkf_ift_amd import evaluate as E
            from adkf_ift_amd.models import ADKTModel, ADKTModelConfig
            from adkf_ift_amd.synthetic import meta_test_tasks
            t0 = time.perf_counter()
            mt_tasks, sizes = meta_test_tasks(args.meta_test_tasks, 64)
            t_gen = time.perf_counter() - t0
            torch.manual_seed(0)
            model = ADKTModel(ADKTModelConfig()).to(dev)
            E.evaluate_tasks(model, mt_tasks[:4], tasks_per_call=4)     # warm-up: first touch of the extractor kernels / allocator
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            res = E.evaluate_tasks(model, mt_tasks, tasks_per_call=16)
            torch.cuda.synchronize(dev)
            mt = time.perf_counter() - t0
            agg = E.avg_metrics_over_tasks({k: [v] for k, v in res.items()})
            meta_test = {"walltime_s": mt, "tasks": len(mt_tasks), "tasks_per_s": len(mt_tasks) / mt, "support": 64,
                         "query_molecules": int(sizes.sum()), "tasks_per_call": 16,
                         "workload": f"{len(mt_tasks)} synthetic molecular tasks, support 64, query sizes {int(sizes.min())}..{int(sizes.max())}, "
                                     "default GNN+ECFP+fc model (random weights), fit to convergence + predict + metrics; clock covers "
                                     "collation, host->device copies, extractor, GP, metrics - not file reading",
                         "synthetic_generation_s": t_gen, "mean_avg_precision": float(agg["avg_precision"][0]),
                         "reference_published_cpu_walltime_s": 121.1,
                         "reference_source": "visualize_results/visualize_classification.ipynb (157 FS-Mol test tasks, support 64, CPU)"}
            del model, mt_tasks
        except Exception as e:   # context only: never a reason to lose the headline line
            print(f"[bench] meta_test not measured ({type(e).__name__}: {e})", file=sys.stderr)

    # ---- the other BASELINE.json configurations, measured after the headline loop and outside its clock (like `converged` and
    # `meta_test`): side objects of the SAME line, so that the driver's record holds every config and not only C2 ----
    side = {}
    is_c2 = (T, N, Nq, d, I) == (256, 128, 128, 256, 20) and args.kernel == "rbf" and not (args.ard or args.regression or args.converge or args.graph)
    want_side = args.side_configs == "on" or (args.side_configs == "auto" and world == 1 and is_c2)
    if rank == 0 and want_side:
        def gp_side(name, T_, N_, d_, steps, warmup, label):
            """One more GP workload through the same meta_step: ms/step, tasks/s, the fit's share of the FP32 peak (HIP events of
            adkf_fit_options_t on the launch stream, like the headline's roofline object)."""
            try:
                w_ = Workload(T_, N_, N_, d_)
                ev_ = make_events(steps)
                dt_, th_ = w_.timed_loop(step_cfg(False), steps, warmup, ev_)
                fms = sum(a.elapsed_time(b_) for a, b_ in ev_) / len(ev_)
                fl_ = roofline.flops_per_task(N_, N_, d_, I)
                ach = fl_["inner_fit"] * T_ / (fms * 1e-3) / 1e12
                side[name] = {"workload": label, "ms_per_step": dt_ / steps * 1e3, "tasks_per_s": T_ * steps / dt_, "steps": steps,
                              "host_enqueue_ms_per_step": th_ / steps * 1e3,
                              "fit": {"avg_ms": fms, "algorithmic_flops": fl_["inner_fit"] * T_, "achieved_tflops": ach,
                                      "frac_of_fp32_peak": ach / roofline.PEAK_FP32_TFLOPS},
                              "whole_path_frac_of_fp32_peak": T_ * steps / dt_ * fl_["total"] / 1e12 / roofline.PEAK_FP32_TFLOPS}
                del w_
            except Exception as e:   # side objects never cost the headline line
                print(f"[bench] side config {name} not measured ({type(e).__name__}: {e})", file=sys.stderr)
            torch.cuda.empty_cache()

        gp_side("c1", 64, 32, 64, 50, 10, f"C1: 64 tasks, N_support = N_query = 32, d = 64, exactly {I} evaluations (host-enqueue-bound: compare host_enqueue_ms_per_step)")
        if "c1" in side:   # the same shape with the GP section replayed from a captured HIP graph (trainer.GraphedGPBackend): what a host-bound shape wants
            try:
                w_ = Workload(64, 32, 32, 64)
                dt_, th_ = w_.timed_loop(step_cfg(False), 50, 10, None, backend=GraphedGPBackend())
                side["c1"]["with_hip_graph"] = {"ms_per_step": dt_ / 50 * 1e3, "tasks_per_s": 64 * 50 / dt_, "host_enqueue_ms_per_step": th_ / 50 * 1e3}
                del w_
            except Exception as e:
                print(f"[bench] c1 with the HIP-graph backend not measured ({type(e).__name__}: {e})", file=sys.stderr)
            torch.cuda.empty_cache()
        gp_side("t512", 512, 128, 256, 20, 5, f"512 tasks/GPU/step at the C2 shape (two tasks per CU in the inner fit), exactly {I} evaluations")
        gp_side("c5", 8, 1024, 512, 5, 2, f"C5: 8 tasks, N_support = N_query = 1024, d = 512, blocked sweep, exactly {I} evaluations (a throughput point: not converged at {I})")
        gp_side("c5_t32", 32, 1024, 512, 3, 1, f"the C5 shape with 32 tasks per step (four per XCD instead of one: the blocked sweep's diagonal chain of one task "
                                                f"runs beside the updates of the others), exactly {I} evaluations")
        if "c5" in side:
            pm, src = profile_json("c5_fit_pmc.json")
            if pm is not None:
                tr = (2.0 * pm["FETCH_SIZE_KB_per_fit"] + pm["WRITE_SIZE_KB_per_fit"]) * 1024.0
                comp = roofline.bytes_per_task(1024, 1024, 512) * 8
                side["c5"]["fit"].update({"traffic": tr, "compulsory_bytes": comp, "traffic_ratio": tr / comp, "traffic_source": src,
                                          "launches_per_fit": pm.get("launches_per_fit")})

        # C3: the full inner loop with the default 25 M-parameter model on synthetic 16-shot molecular tasks (tools/bench_c3.py is the
        # stand-alone form), plus the same step at the reference's CLI shape (support 64, query 256: fs_mol/adaptive_dkt_train.py:50-61)
        def c3_side(name, n_tasks, support, query, steps, warmup, with_cpu):
            try:
                from adkf_ift_amd.meta_batch import collate_meta_batch, model_meta_step
                from adkf_ift_amd.models import ADKTModel, ADKTModelConfig
                from adkf_ift_amd.synthetic import molecular_task
                gen = torch.Generator().manual_seed(0)
                mtasks = [molecular_task(support, query, gen) for _ in range(n_tasks)]
                mb = collate_meta_batch(mtasks).to(dev)
                torch.manual_seed(0)
                model = ADKTModel(ADKTModelConfig()).to(dev)   # reference defaults: gnn+ecfp+fc, Matern-5/2, 2048-d features
                opt_ = torch.optim.Adam(model.feature_extractor_params(), lr=1e-4, fused=True)
                mcfg = MetaStepConfig(gp_kernel="matern", clip_value=1.0, inner_max_evals=200)
                for _ in range(warmup):
                    model_meta_step(model, opt_, mb, mcfg)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(steps):
                    model_meta_step(model, opt_, mb, mcfg)
                torch.cuda.synchronize(dev)
                dt_ = (time.perf_counter() - t0) / steps
                V = int(mb.molecules.node_features.shape[0])
                E = 2 * sum(int(a.shape[0]) for a in mb.molecules.adjacency_lists)
                G = int(mb.molecules.num_graphs)
                fl_ = roofline.dense_flops_c3(V, E, G)
                gp_fl = sum(roofline.flops_per_task(support, query, 2048, 20)["total"] for _ in range(n_tasks))
                ach = fl_["step"] / dt_ / 1e12
                obj = {"workload": f"C3: {n_tasks} tasks/step, {support}-shot, {query} query molecules per task, default GNN+ECFP+fc model "
                                   f"({sum(p.numel() for p in model.parameters()) / 1e6:.1f} M parameters, random weights), Matern-5/2, inner fit to "
                                   "convergence, one extractor forward / backward per meta-batch, Adam + clip 1.0",
                       "ms_per_step": dt_ * 1e3, "tasks_per_s": n_tasks / dt_, "steps": steps, "nodes": V, "message_edges": E, "molecules": G,
                       "roofline": {"bound": "mfma", "model": "dense layers of the extractor and head, 3 x forward (adkf_ift_amd/roofline.py::dense_flops_c3); "
                                                               "the GP section's FLOPs are listed beside it, not added",
                                    "dense_flops_per_step": fl_["step"], "gp_flops_per_step_at_20_evals": gp_fl, "achieved": ach,
                                    "peak": roofline.PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": ach / roofline.PEAK_FP32_TFLOPS}}
                del model, opt_, mb
                torch.cuda.empty_cache()
                if with_cpu and not args.no_cpu_baseline:
                    from oracle import gp_oracle as O_
                    from oracle import ref_cpu_path
                    torch.manual_seed(0)
                    cpu_model = ADKTModel(ADKTModelConfig())
                    rate, n_done, cores, nfev = ref_cpu_path.time_model_tasks(cpu_model, mtasks, O_.KERNEL_MATERN52, budget_s=min(args.cpu_baseline_seconds, 10.0))
                    obj["cpu_baseline"] = {"value": rate, "unit": "tasks/s", "cores": cores, "kind": "port",
                                           "sample": f"first {n_done} task(s) of the same meta-batch, sequential, the reference algorithm through the whole model in "
                                                     f"float32 PyTorch on the host (per task: SciPy L-BFGS-B fit, mean {nfev:.0f} evals; dense Hessian + nested-Jacobian "
                                                     f"hypergradient = 3 double-backward passes through the extractor), torch threads = {cores}"}
                    del cpu_model
                side[name] = obj
            except Exception as e:
                print(f"[bench] side config {name} not measured ({type(e).__name__}: {e})", file=sys.stderr)
            torch.cuda.empty_cache()

        c3_side("c3", 16, 16, 128, 4, 2, True)
        c3_side("c3_cli_defaults", 16, 64, 256, 3, 1, False)

    if rank == 0:
        fl = roofline.flops_per_task(N, Nq, d, I)
        total_tasks = T * world * args.steps
        value = total_tasks / dt
        cfg_name = {(256, 128, 256): "C2", (64, 32, 64): "C1", (8, 1024, 512): "C5"}.get((T, N, d), "custom")
        if args.global_tasks == 512 and (N, d) == (128, 256):
            cfg_name = "C4"
        if args.ard:
            cfg_name += " with the ARD kernel (roofline FLOP model below is the non-ARD one: indicative only)"
        roof = None
        if fit_ms is not None:
            fit_flops = fl["inner_fit"] * T            # algorithmic FLOPs of ONE launch of the dominant kernel
            achieved = fit_flops / (fit_ms * 1e-3) / 1e12
            # HBM bytes of that launch: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE in separate passes of this same
            # command (tools/profile_round.sh), committed under profiles/ with the commit they were taken at; FETCH_SIZE is
            # doubled as MI355X_MICROARCH.md (HBM section) prescribes for 16-byte-per-lane streaming reads on gfx950.
            traffic = traffic_src = None
            # the <= 128-register build (two tasks per CU) is taken when the batch has more tasks than the chip has CUs (adkf_gp.hip: num_cus())
            low = T > torch.cuda.get_device_properties(dev).multi_processor_count
            plain = args.kernel == "rbf" and not args.ard and not args.regression
            pm, pm_file = profile_json("k_inner_pmc.json")
            if pm is not None and (T, N, Nq, d, I) == (256, 128, 128, 256, 20) and plain:
                traffic = (2.0 * pm["FETCH_SIZE_KB_per_launch"] + pm["WRITE_SIZE_KB_per_launch"]) * 1024.0
                traffic_src = {"file": pm_file, "commit": pm.get("commit"), "correction": "2 x FETCH_SIZE + WRITE_SIZE"}
            pm, pm_file = profile_json("c5_fit_pmc.json")
            if pm is not None and (T, N, Nq, d, I) == (8, 1024, 1024, 512, 20) and plain:
                traffic = (2.0 * pm["FETCH_SIZE_KB_per_fit"] + pm["WRITE_SIZE_KB_per_fit"]) * 1024.0
                traffic_src = {"file": pm_file, "commit": pm.get("commit"), "launches_per_fit": pm.get("launches_per_fit"),
                               "compulsory_bytes": roofline.bytes_per_task(N, Nq, d) * T,
                               "correction": "2 x FETCH_SIZE + WRITE_SIZE, summed over the launches of one adkf_fit call (by dispatch order)"}
            if args.ard:
                fit_kernel, bound = "ARD inner fit (all launches between the two events)", "mfma"
            elif N <= 128:
                fit_kernel = ("k_inner (in-kernel quasi-Newton fit: kernel build + register-resident sweep per evaluation; the sweep's "
                              "rank-4 updates run as v_mfma_f32_16x16x4_f32" + ("; <= 128-register build, two tasks per CU)" if low and N > 64 else ")"))
                bound = "mfma"   # FP32 matrix pipe (157.3 TFLOP/s = the FP32 vector peak); the kernel is limited by the hand-off chain of the sweep
            else:
                fit_kernel = ("blocked inner fit (all launches between the two events: k_lg_build, k_lg_diag, panel/update "
                              "MFMA GEMMs, k_lg_traces, k_lg_advance per evaluation)")
                bound = "mfma"
            roof = {"kernel": fit_kernel, "bound": bound, "achieved": achieved, "peak": roofline.PEAK_FP32_TFLOPS,
                    "unit": "TFLOP/s", "frac": achieved / roofline.PEAK_FP32_TFLOPS, "traffic": traffic,
                    "traffic_source": traffic_src, "flops_per_launch": fit_flops, "avg_launch_ms": fit_ms}
        dz_exec = 2 * d * (N * N + 2 * N * Nq + Nq * Nq)   # what ProbDZ executes (K = N + Nq per output row)
        line = {
            "metric": metric_name(N, d), "value": value, "unit": "tasks/s", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True,
            "scaling": "strong" if args.global_tasks else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{cfg_name}{' (regression labels, noise 0.01)' if args.regression else ''}: {T} tasks/GPU/step, N_support={N}, N_query={Nq}, d={d}, kernel={args.kernel}, "
                                   f"inner fit = {'to convergence' if args.converge else f'exactly {I} MLL value+grad evals'}, "
                                   "IFT hypergradient, theta = W[d,d] linear feature map, Adam + clip 1.0",
                       "tasks_per_gpu": T, "global_tasks": T * world, "parallelism": f"task-sharded dp{world}",
                       "library_gemm_choice": args.gemm_tuning},
            "whole_path_tflops": value * fl["total"] / 1e12,
            "whole_path_frac_of_fp32_peak": value * fl["total"] / 1e12 / (roofline.PEAK_FP32_TFLOPS * world),
            # the frozen SURVEY 8d model prices the dZ GEMMs at 4d(N^2+N Nq+Nq^2); the kernels execute 2d(N+Nq)^2
            "whole_path_frac_executed_flops": value * (fl["total"] - fl["dZ"] + dz_exec) / 1e12 / (roofline.PEAK_FP32_TFLOPS * world),
            "roofline": roof,
            "converged": converged,
            "meta_test": meta_test,
            "host_enqueue_ms_per_step": t_host / args.steps * 1e3,
            "cpu_baseline": cpu_baseline,
            "cpu_baseline_twin": cpu_twin,
            "parity": parity,
        }
        # second roofline entry: the outer / hypergradient kernel of the C2 step, from this round's rocprofv3 runs of this command
        if is_c2 and world == 1:
            hy, hy_src = profile_json("k_hyper_pmc.json")
            if hy is not None and hy.get("avg_duration_us"):
                hf = 13 * N ** 3 * T
                ach = hf / (hy["avg_duration_us"] * 1e-6) / 1e12
                tr = (2.0 * hy["FETCH_SIZE_KB_per_launch"] + hy["WRITE_SIZE_KB_per_launch"]) * 1024.0
                line["roofline_k_hyper"] = {"kernel": "k_hyper<true,0> (the whole outer / hypergradient stage of a task in one workgroup)", "bound": "mfma",
                                            "achieved": ach, "peak": roofline.PEAK_FP32_TFLOPS, "unit": "TFLOP/s", "frac": ach / roofline.PEAK_FP32_TFLOPS,
                                            "traffic": tr, "compulsory_bytes": 7 * N * N * 4 * T, "traffic_ratio": tr / (7 * N * N * 4 * T),
                                            "flops_per_launch": hf, "avg_launch_ms": hy["avg_duration_us"] * 1e-3,
                                            "source": hy_src + " (rocprofv3 --kernel-trace average and the two --pmc passes of this command; not measured live)"}
        line.update(side)
        print(json.dumps(line), flush=True)
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
