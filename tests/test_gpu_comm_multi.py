"""The RCCL transport of the multi-device C ABI with MORE THAN ONE RANK: one process per GPU, as deployed.
Runs only where at least two GPUs are visible (the round's gpurun boxes have one: skipped there; the driver's 8-GPU node and any
multi-GPU host execute it).  tests/comm_multi_worker.py is the per-rank program; this test launches it under
torch.distributed.run and requires every rank to exit 0."""
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _device_count():
    import torch
    return torch.cuda.device_count() if torch.cuda.is_available() else 0


@pytest.mark.parametrize("world", [1, 2, 4, 8])      # 1: the worker itself, on any GPU box
def test_rccl_communicator_of_several_processes(world):
    if _device_count() < world:
        pytest.skip("needs %d GPUs, %d visible" % (world, _device_count()))
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "comm_multi_worker.py")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    out = r.stdout.decode(errors="replace")
    assert r.returncode == 0, out[-4000:]
    for k in range(world):
        assert "rank %d of %d: ok" % (k, world) in out, out[-4000:]
