"""(Named to run LAST: `pytest -x` stops at the first failure, and a multi-GPU rendezvous is the one test here that depends on
the node around the GPU.)

GPU, two ranks over RCCL (backend "nccl"): the collective of SURVEY section 8(e) on real hardware.  Skipped on a one-GPU
box (the driver's multi-GPU node runs it).  The ranks are FRESH processes started with subprocess - this pytest process has
already initialised the GPU and must not fork into, or exec, GPU work.

Reference counterpart of the reduction: fs_mol/utils/adaptive_dkt_utils.py:356,402-410 (task-mean of the hypergradients,
clip-by-global-norm AFTER it, one optimiser step); the expected numbers are the harness fixture produced with the
reference's cauchy_hypergradient (tests/golden/make_golden.py)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
TOL = 1e-4


def _need_two_gpus():
    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs (RCCL over xGMI); this box has %d" % torch.cuda.device_count())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _rel(a, ref):
    a, ref = np.asarray(a, dtype=np.float64), np.asarray(ref, dtype=np.float64)
    return np.abs(a - ref).max() / max(np.abs(ref).max(), 1e-30)


def _run_ranks(argv, world, backend, timeout=600):
    port = str(_free_port())
    procs = []
    for r in range(world):
        env = dict(os.environ, ADKF_TEST_DIST_BACKEND=backend, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=port, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        procs.append(subprocess.Popen([sys.executable] + argv, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT))
    outs = []
    try:
        for p in procs:
            out, _ = p.communicate(timeout=timeout)
            outs.append(out.decode(errors="replace"))
    finally:
        for p in procs:           # the exact children started above, never a pattern
            if p.poll() is None:
                p.kill()
    for r, (p, out) in enumerate(zip(procs, outs)):
        assert p.returncode == 0, f"rank {r} exited with {p.returncode}:\n{out[-3000:]}"
    return outs


@pytest.mark.parametrize("backend", ["nccl", "gloo"])
@pytest.mark.parametrize("split,stated_total", [((2, 2), None), ((3, 1), None), ((3, 1), 4)])
def test_two_ranks_reproduce_the_reference_step(golden_dir, tmp_path, split, stated_total, backend):
    """backend "nccl": two GPUs, the gradient all-reduce over RCCL / xGMI (skipped on a one-GPU box).  backend "gloo": the SAME
    rank script with both ranks on cuda:0 and the reduction through the host - runs everywhere, so the script, the sharding
    and the fixture comparison are exercised on every box and only the transport is left to the two-GPU run."""
    if backend == "nccl":
        _need_two_gpus()
    fixture = os.path.join(golden_dir, "harness_T4_N128_d256_k0.npz")
    g = np.load(fixture)
    prefix = str(tmp_path / "nccl")
    argv = [os.path.join(HERE, "_nccl_rank.py"), fixture, prefix, ",".join(map(str, split))]
    if stated_total is not None:
        argv.append(str(stated_total))
    _run_ranks(argv, 2, backend)
    res = [np.load(f"{prefix}_rank{r}.npz") for r in range(2)]
    for r in range(2):
        assert str(res[r]["backend"]) == backend
        # after the all-reduce, / T and the clip every rank holds the reference's clipped task-mean gradient and has stepped by it
        assert _rel(res[r]["grad"], g["grad_clipped"]) <= TOL, (r, _rel(res[r]["grad"], g["grad_clipped"]))
        assert _rel(res[r]["step"], g["grad_clipped"]) <= TOL
        lo = sum(split[:r])
        assert _rel(res[r]["losses"] * int(g["N"]), g["f_out"][lo:lo + split[r]]) <= TOL
    assert np.array_equal(res[0]["grad"], res[1]["grad"])        # bit-identical replicas: the step must not drift apart


def test_bench_two_gpus_prints_one_line_for_two_ranks():
    _need_two_gpus()
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--global-tasks", "512", "--steps", "5",
                          "--warmup", "2", "--no-cpu-baseline", "--converge-steps", "0"], capture_output=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr.decode(errors="replace")[-3000:]
    line = json.loads(out.stdout.decode().strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["value"] > 0
    assert line["config"]["tasks_per_gpu"] == 256
