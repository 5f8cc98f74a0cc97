"""bench.py's own launcher (reduction it times: fs_mol/utils/adaptive_dkt_utils.py:402-410):
``python bench.py --gpus N`` without WORLD_SIZE starts N rank processes itself and relays rank 0's line.  Exercised
here without a GPU through ``--dry-run`` (ranks rendezvous over gloo on 127.0.0.1 and all-reduce a dummy gradient; no GP
arithmetic is performed or faked - ``value`` is null)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**extra):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR")}
    env.update(ADKF_BENCH_BACKEND="gloo", **extra)
    return env


def _line(stdout: str) -> dict:
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


@pytest.mark.parametrize("gpus", [1, 2])
def test_gpus_flag_launches_that_many_ranks(gpus):
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(gpus), "--steps", "3", "--dry-run"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    line = _line(r.stdout)
    assert line["n_gpus"] == gpus and line["dry_run"] is True and line["value"] is None
    assert line["scaling"] == "weak" and line["config"]["tasks_per_gpu"] == 256
    assert line["config"]["parallelism"] == f"task-sharded dp{gpus}"


def test_strong_scaling_splits_the_global_batch():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--global-tasks", "512", "--steps", "2", "--dry-run"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    line = _line(r.stdout)
    assert line["n_gpus"] == 2 and line["scaling"] == "strong" and line["config"]["tasks_per_gpu"] == 256


def test_mismatch_between_gpus_and_world_size_is_refused():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--dry-run"], env=_env(WORLD_SIZE="1", RANK="0"),
                       capture_output=True, text=True, timeout=120)
    assert r.returncode != 0 and "WORLD_SIZE=1" in r.stderr


def test_a_dead_rank_ends_the_run_instead_of_hanging_it():
    """Rank 1 exits before the rendezvous: rank 0 would wait for it for ever - the parent must notice, end rank 0 and return
    the dead rank's status."""
    import time
    t0 = time.monotonic()
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "2", "--dry-run"], env=_env(ADKF_BENCH_DRYRUN_FAIL_RANK="1"),
                       capture_output=True, text=True, timeout=240)
    assert r.returncode == 3, (r.returncode, r.stderr[-2000:])
    assert time.monotonic() - t0 < 200


def test_metric_string_follows_the_configuration():
    sys.path.insert(0, ROOT)
    import bench
    assert bench.metric_name(128, 256) == "meta-tasks/sec (N_support=128, d=256)"
    assert bench.metric_name(1024, 512) == "meta-tasks/sec (N_support=1024, d=512)"
