import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

P = 2**64 - 2**32 + 1


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: test needs a real MI355X (run with -m gpu via gpurun)")
    _guarded_session()


def _guarded_session():
    """WF_DEBUG_GUARD=1 (or 2): the electric-fence session.  Every device allocation of the library AND — through torch's pluggable
    allocator, pointed at the library's wf_debug_torch_malloc / _free — every tensor the tests hand to it is a block of its own with
    unmapped pages behind its last byte (WF_DEBUG_GUARD_ALIGN=left: before its first), so a kernel that steps outside ANY buffer is a
    GPU page fault at that instruction, with the test's name on the screen (-v) and HIP_LAUNCH_BLOCKING=1 naming the call.  Mode 2:
    red zones checked at free time instead (no virtual-memory API needed).  tools/guard_session.sh is the command line."""
    if os.environ.get("WF_DEBUG_GUARD", "0") in ("", "0"):
        return
    import torch
    lib = os.path.join(ROOT, "winterfell_amd", "libwinterfell_hip.so")
    alloc = torch.cuda.memory.CUDAPluggableAllocator(lib, "wf_debug_torch_malloc", "wf_debug_torch_free")
    torch.cuda.memory.change_current_allocator(alloc)


@pytest.fixture(scope="session")
def golden():
    with open(os.path.join(ROOT, "tests", "golden", "reference_vectors.json")) as f:
        return json.load(f)


# OpenMP team of the oracle's `par` paths in the tests.  NOT the host's thread count: the GPU boxes show 256 hardware threads but run
# under a CPU quota, and a 256-thread team that spins at every barrier burns the quota and is throttled for the rest of each
# scheduler period — round 6 measured ~90 ms per parallel region: a 2^18 x 32 commitment took 115 s of oracle time, a 2^12-row one 106 s,
# ~700 s of a 1010 s suite.  (bench.py's cpu_baseline scans team sizes and reports the best; this constant is test plumbing only.)
ORACLE_THREADS = max(1, min(16, os.cpu_count() or 1))
os.environ.setdefault("OMP_WAIT_POLICY", "PASSIVE")      # read by libgomp when the oracle library loads: idle team members sleep


@pytest.fixture(scope="session")
def oracle():
    import oracle as orc
    orc.build()
    orc.set_num_threads(ORACLE_THREADS)
    return orc


def splitmix64(seed, n):
    """SURVEY.md section 8(d) input generator: SplitMix64(seed) -> rejection-sample < p (canonical ints)."""
    out = np.empty(n, dtype=np.uint64)
    x = seed & 0xFFFFFFFFFFFFFFFF
    i = 0
    mask = 0xFFFFFFFFFFFFFFFF
    while i < n:
        x = (x + 0x9E3779B97F4A7C15) & mask
        z = x
        z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & mask
        z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & mask
        z = z ^ (z >> 31)
        if z < P:
            out[i] = z
            i += 1
    return out


def rand_field(seed, n):
    """Fast uniform canonical field elements (numpy PCG), for large inputs."""
    rng = np.random.default_rng(seed)
    v = rng.integers(0, P, size=n, dtype=np.uint64)
    return v
