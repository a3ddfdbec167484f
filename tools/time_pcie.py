"""PCIe-inclusive rates of the boundary when the caller hands over HOST buffers: H2D of a 2^24 x 8 B
vector and D2H of the result through wf_memcpy_*, pageable vs page-locked with wf_host_register, next to the NTT itself.
Also the 2^20 x 4 trace -> (polys, LDE, nodes) round trip of wf_build_trace_commitment."""
import ctypes
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import winterfell_amd
from winterfell_amd.math import fft

ctx = winterfell_amd.default_context(0)
vp = lambda a: a.ctypes.data_as(ctypes.c_void_p)


def timed(fn, reps=5):
    fn()
    ts = []
    for _ in range(reps):
        ctx.sync()
        t = time.perf_counter()
        fn()
        ctx.sync()
        ts.append((time.perf_counter() - t) * 1e3)
    return float(np.median(ts))


for log_n in (24,):
    n = 1 << log_n
    host = np.random.default_rng(1).integers(0, 1 << 63, n, dtype=np.uint64)
    back = np.empty_like(host)
    dev = ctx.empty_u64(n)
    dp = ctypes.c_void_p(dev.data_ptr())
    h2d = lambda: ctx.call("wf_memcpy_h2d", dp, vp(host), host.nbytes)
    d2h = lambda: ctx.call("wf_memcpy_d2h", vp(back), dp, host.nbytes)
    ntt = lambda: fft.evaluate_poly(dev, ctx=ctx)
    res = {"pageable_h2d_ms": timed(h2d), "pageable_d2h_ms": timed(d2h)}
    ctx.call("wf_host_register", vp(host), host.nbytes)
    ctx.call("wf_host_register", vp(back), back.nbytes)
    res.update({"registered_h2d_ms": timed(h2d), "registered_d2h_ms": timed(d2h), "ntt_ms": timed(ntt)})
    res["registered_roundtrip_ms"] = timed(lambda: (h2d(), ntt(), d2h()))
    ctx.call("wf_host_unregister", vp(host))
    ctx.call("wf_host_unregister", vp(back))
    gb = host.nbytes / 1e9
    print("2^%d f64 vector (%.0f MB):" % (log_n, host.nbytes / 1e6), {k: round(v, 3) for k, v in res.items()},
          "| GB/s: pageable h2d %.1f d2h %.1f, registered h2d %.1f d2h %.1f" % (
              gb / res["pageable_h2d_ms"] * 1e3, gb / res["pageable_d2h_ms"] * 1e3, gb / res["registered_h2d_ms"] * 1e3,
              gb / res["registered_d2h_ms"] * 1e3),
          "| elements/s incl. PCIe (registered, h2d + ntt + d2h): %.3g" % (n / res["registered_roundtrip_ms"] * 1e3))
