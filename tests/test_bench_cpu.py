"""bench.py's multi-rank plumbing without a GPU: `--gpus 2` with no launcher in the environment must start two ranks
itself and rank 0 must print ONE JSON line saying n_gpus = 2 (the flag used to be parsed and ignored).
The gloo backend swaps the engine for a CPU stub -- the line says so and carries no value."""
import json
import os
import subprocess
import sys

from conftest import ROOT


def _run(extra, env_extra=None, timeout=300):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra or {})
    return subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + extra, env=env, stdout=subprocess.PIPE,
                          stderr=subprocess.PIPE, timeout=timeout, cwd=ROOT)


def test_gpus_2_spawns_two_ranks_and_reports_them():
    r = _run(["--gpus", "2", "--backend", "gloo", "--steps", "3", "--warmup", "1"])
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    lines = [ln for ln in r.stdout.decode().splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 3 and out["warmup"] == 1
    assert out["stub"] is True and out["value"] is None          # not a measurement
    assert out["extra"]["allreduce_check"] is True                # the gradient all-reduce summed over both ranks


def test_more_gpus_than_devices_is_an_error_not_a_single_rank_run():
    r = _run(["--gpus", "8", "--steps", "1", "--warmup", "0"], timeout=120)
    assert r.returncode != 0
    assert b"device(s) visible" in r.stderr and r.stdout.strip() == b""
