#!/bin/bash
# The electric-fence session (tests/conftest.py _guarded_session, csrc/context.hip guard_alloc): the GPU suite with every device
# buffer — the library's and the tests' — ending on the last mapped byte of its own address range.  xdist (-n 1) restarts the
# worker when a test takes the process down with a GPU fault, so one run lists every faulting test instead of stopping at the first.
#   tools/guard_session.sh [pytest args]        output: gpurun_out/guard_session.log
cd "$(dirname "$0")/.." || exit 1
mkdir -p gpurun_out
export WF_DEBUG_GUARD=${WF_DEBUG_GUARD:-1} HIP_LAUNCH_BLOCKING=1 AMD_SERIALIZE_KERNEL=3
# does the virtual-memory API work here?  (an allocation, a copy each way and a transform under guard pages) — else red zones
if [ "$WF_DEBUG_GUARD" = 1 ] && ! python - > gpurun_out/guard_probe.log 2>&1 <<'PY'
import os, sys
sys.path.insert(0, os.getcwd())
sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import conftest
conftest._guarded_session()
import numpy as np, torch
import winterfell_amd
from winterfell_amd.math import fft
ctx = winterfell_amd.default_context()
assert ctx.lib.wf_debug_guard_mode() == 1
a = np.arange(1 << 12, dtype=np.uint64)
b = fft.interpolate_poly(fft.evaluate_poly(a.copy()))
assert np.array_equal(a, b)
print("guard pages ok")
PY
then
    echo "guard pages unavailable (gpurun_out/guard_probe.log): falling back to red zones" | tee -a gpurun_out/guard_probe.log
    export WF_DEBUG_GUARD=2
fi
echo "WF_DEBUG_GUARD=$WF_DEBUG_GUARD align=${WF_DEBUG_GUARD_ALIGN:-right}" > gpurun_out/guard_session.log
python -m pytest tests -m gpu -v -p no:cacheprovider -n 1 --max-worker-restart=40 "$@" >> gpurun_out/guard_session.log 2>&1
rc=$?
grep -E "^=+ .*(passed|failed)|crashed|Memory access fault|FAILED|ERROR" gpurun_out/guard_session.log | tail -40
exit $rc
