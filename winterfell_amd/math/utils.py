"""math::utils on the GPU (math/src/utils/mod.rs): the series and batch-inversion helpers the hot-path functions are built
from.  Base-field vectors in internal form; numpy in -> numpy out, device tensor in -> device tensor out."""
import ctypes

import numpy as np

from .._lib import default_context, ptr
from . import fields


def get_power_series_with_offset(b, s, n, ctx=None, field=fields.f64):
    """get_power_series_with_offset (utils/mod.rs:69-79): [s, s*b, s*b^2, ...] (n elements, device tensor).  b, s: python
    ints in internal form (e.g. field.new(3))."""
    ctx = ctx or default_context()
    out = ctx.empty_u64(max(n, 1) * field.W)[: n * field.W]
    bw, sw = field.element_words(int(b)), field.element_words(int(s))
    ctx.call("wf_get_power_series_with_offset", field.ID, bw.ctypes.data_as(ctypes.c_void_p), sw.ctypes.data_as(ctypes.c_void_p), n,
             ptr(out) if n else None)
    return out


def get_power_series(b, n, ctx=None, field=fields.f64):
    """get_power_series (utils/mod.rs:36-46)."""
    return get_power_series_with_offset(b, field.new(1), n, ctx, field)


def batch_inversion(values, ctx=None, field=fields.f64):
    """batch_inversion (utils/mod.rs:169-215): element-wise inverses, zeros stay zero."""
    ctx = ctx or default_context()
    host = isinstance(values, np.ndarray)
    d = ctx.to_device(values) if host else values
    n = d.numel() // field.W
    out = ctx.empty_u64(max(n, 1) * field.W)[: n * field.W]
    ctx.call("wf_batch_inversion", field.ID, ptr(d) if n else None, n, ptr(out) if n else None)
    return ctx.to_host(out) if host else out
