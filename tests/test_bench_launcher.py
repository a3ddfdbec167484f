"""bench.py's launcher contract on a box without GPUs: `--gpus N` spawns N ranks itself (torch.distributed.run on 127.0.0.1),
refuses to report an N-GPU number on fewer devices, and rejects a launcher / --gpus mismatch.  The N > 1 control flow
(rendezvous, barriers around the timed region, MAX over ranks, one JSON line from rank 0) runs here over gloo in --dry-run
mode; the measured configuration is always nccl (= RCCL) with one device per rank."""
import json
import os
import subprocess
import sys

import pytest

from conftest import ROOT

BENCH = os.path.join(ROOT, "bench.py")


def _run(args, env=None, timeout=300):
    e = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        e.pop(k, None)
    e.update(env or {})
    return subprocess.run([sys.executable, BENCH] + args, capture_output=True, text=True, timeout=timeout, env=e, cwd=ROOT)


def _json_line(stdout):
    lines = [l for l in stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, stdout
    return json.loads(lines[0])


def test_gpus_flag_spawns_that_many_ranks_over_gloo():
    r = _run(["--gpus", "2", "--dry-run"], {"WF_BENCH_BACKEND": "gloo"})
    assert r.returncode == 0, r.stderr[-2000:]
    out = _json_line(r.stdout)
    assert out["n_gpus"] == 2 and out["dry_run"] is True and out["backend"] == "gloo"


def test_single_rank_dry_run_prints_one_line():
    r = _run(["--dry-run"])
    assert r.returncode == 0, r.stderr[-2000:]
    assert _json_line(r.stdout)["n_gpus"] == 1


def test_refuses_more_ranks_than_devices():
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this box really has 2 devices")
    r = _run(["--gpus", "2"])
    assert r.returncode != 0
    assert "HIP device" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


def test_rejects_launcher_mismatch():
    # torchrun started 1 rank but the command line says 2 GPUs: never silently measure the smaller job
    r = _run(["--gpus", "2", "--dry-run"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0", "WF_BENCH_BACKEND": "gloo"})
    assert r.returncode != 0 and "one rank per GPU" in r.stderr
