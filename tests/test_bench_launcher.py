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


# ---- the ONE stdout line stays small (round 5: a 20 KB line came back from the driver as `parsed: null`) ----

def _full_size_detail():
    """a full-size result object: the committed round-5 line (every roofline case, counters, per-thread CPU tables: 20 KB), plus the
    N > 1 legs' keys, so the serialisation is exercised on what a real run produces"""
    with open(os.path.join(ROOT, "profiles", "r05", "bench_driver_protocol.json")) as f:
        out = json.load(f)
    assert len(json.dumps(out)) > 16000
    return out


def test_compact_line_fits_the_driver_and_round_trips():
    sys.path.insert(0, ROOT)
    import bench
    out = _full_size_detail()
    line = bench.compact_line(out, "gpurun_out/bench_detail_n1.json")
    assert "\n" not in line and len(line) < 6144 == bench.LINE_LIMIT
    got = json.loads(line)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config"):
        assert got[k] == out[k], k
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert got["roofline"][k] == out["roofline"][k], k
    assert got["roofline"]["frac"] == pytest.approx(got["roofline"]["achieved"] / got["roofline"]["peak"])
    for k in ("value", "unit", "cores", "kind", "sample"):
        assert k in got["cpu_baseline"]
    assert got["cpu_baseline"]["value"] == out["cpu_baseline"]["value"]
    assert 0 < len(got["extra"]) <= 10 and all(not isinstance(v, (dict, list)) for v in got["extra"].values())
    assert all(not isinstance(v, (dict, list)) for v in got["roofline"].values())           # scalars only: nothing that can grow
    assert set(got["roofline_frac_by_case"]) == {k for k, v in out["rooflines"].items() if "frac" in v}


def test_compact_line_of_a_multi_gpu_run_and_the_limit():
    sys.path.insert(0, ROOT)
    import bench
    out = _full_size_detail()
    out["n_gpus"] = 8
    for pre in ("config3_sharded_commit_f128_2^22x64_b8_blake3_p8", "config4_sharded_fri_2^24_quad_fold4_blake3_n8"):
        out["extra"].update({pre + "_ms": 1.5, pre + "_rank0_kernel_ms": 1.0, pre + "_exchanged_bytes_per_rank": 1 << 30, pre + "_roots_agree": True,
                             pre + "_root": "ab" * 32})
    out["extra"].update({"sharded_lde_commit_ms_2^20x32_b8_blake3": 1.0, "merkle_blake3_leaves_per_s_2^23_all_ranks": 3e11,
                         "strided_lde_commit_ms_2^20x4_b8_blake3": 2.0, "sharded_fri_build_layers_ms_2^24_quad_fold4_blake3": 3.0,
                         "partitioned_fri_build_layers_ms_2^24_quad_fold4_blake3": 1.0, "transport": "RCCL via wf_comm_init_rank, 8 ranks"})
    got = json.loads(bench.compact_line(out, None))
    assert len(got["extra"]) == 10 and "config3_sharded_commit_f128_2^22x64_b8_blake3_p8_ms" in got["extra"]
    assert "merkle_blake3_leaves_per_s_2^23_all_ranks" in got["extra"] and "detail" not in got
    # a line that would not fit is an error, never a silently truncated record
    out["config"]["workload"] = "x" * 7000
    with pytest.raises(ValueError):
        bench.compact_line(out, None)


def test_detail_file_holds_everything(tmp_path):
    sys.path.insert(0, ROOT)
    import bench
    out = _full_size_detail()
    p = str(tmp_path / "sub" / "detail.json")
    assert bench.write_detail(out, p) == p
    with open(p) as f:
        assert json.load(f) == out
    assert bench.write_detail(out, "/proc/nope/detail.json") is None                 # an unwritable place never fails the bench
