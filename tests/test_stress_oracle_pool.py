"""CPU: the worker side of tests/test_gpu_stress.py (tests/_stress_oracle.py) runs in spawned CPU-only processes and returns
what the GPU suite compares against - checked here on one small task at the oracle's own optimum."""
import multiprocessing as mp
from concurrent.futures import ProcessPoolExecutor

import numpy as np
import torch

from adkf_ift_amd.synthetic import make_tasks
from oracle import gp_oracle as O
from _stress_oracle import oracle_bundle, worker_init


def test_oracle_bundle_in_a_spawned_pool():
    tasks = make_tasks(2, 12, 5, N_q=7, regression=True, first_task=300)
    Zs, Zq = tasks.features()
    p0, pri = O.init_phi(Zs[0].double(), True, True)
    phi = O.fit_phi(Zs[0].double(), tasks.y_s[0].double(), p0, pri, 0)[0].float()
    with ProcessPoolExecutor(max_workers=2, mp_context=mp.get_context("spawn"), initializer=worker_init) as pool:
        futs = [pool.submit(oracle_bundle, (Zs[t], tasks.y_s[t], Zq[t], tasks.y_q[t], phi, 0, True)) for t in range(2)]
        outs = [f.result(timeout=300) for f in futs]
    o = outs[0]
    assert set(o["q"]) == {"f_in", "H", "f_out", "g_out", "v", "dZs_total", "dZq_total", "pred_mean", "pred_var"}
    assert o["q"]["dZs_total"].shape == (12, 5) and o["q"]["pred_mean"].shape == (7,)
    assert abs(float(o["q"]["f_in"]) - o["f_star"]) <= 1e-5 * abs(o["f_star"])      # phi IS the oracle's optimum (rounded to float32)
    assert np.abs(o["p0"] - p0.numpy()).max() == 0.0 and o["cond"] >= 1.0
    assert all(0.0 <= e < 1e-2 for e in o["e32"].values()) and all(s >= 1.0 for s in o["slack"].values())
