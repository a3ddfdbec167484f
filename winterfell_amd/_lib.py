"""ctypes binding of libwinterfell_hip.so + device-buffer plumbing (torch tensors hold HBM allocations)."""
import atexit
import ctypes
import os
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("WF_HIP_LIBRARY", os.path.join(_HERE, "libwinterfell_hip.so"))   # override: A/B-testing kernel variants

WF_FIELD_F64, WF_FIELD_F128, WF_FIELD_F62 = 0, 1, 2
WF_HASH_BLAKE3_256, WF_HASH_RP64_256, WF_HASH_SHA3_256, WF_HASH_RPJIVE64_256, WF_HASH_RP62_248 = 0, 1, 2, 3, 4
WF_HASH_BLAKE3_192 = 5

_u32, _u64, _int, _vp = ctypes.c_uint32, ctypes.c_uint64, ctypes.c_int, ctypes.c_void_p

# name -> argtypes, exactly the prototypes of include/winterfell_hip.h (all return int unless noted)
_PROTOS = {
    "wf_device_count": [ctypes.POINTER(_int)],
    "wf_ctx_create": [_int, ctypes.POINTER(_vp)],
    "wf_ctx_create_on_stream": [_int, _vp, ctypes.POINTER(_vp)],
    "wf_ctx_destroy": [_vp],
    "wf_ctx_set_stream": [_vp, _vp],
    "wf_ctx_get_stream": [_vp, ctypes.POINTER(_vp)],
    "wf_ctx_sync": [_vp],
    "wf_last_hip_error": [_vp],
    "wf_last_device_status": [_vp],
    "wf_debug_shader_clock": [_vp, _u32, ctypes.POINTER(ctypes.c_double)],
    "wf_debug_poke_tree_ticket": [_vp, _u32],
    "wf_prof_enable": [_vp, _int],
    "wf_prof_collect": [_vp, ctypes.c_char_p, ctypes.c_size_t],
    "wf_malloc": [_vp, ctypes.c_size_t, ctypes.POINTER(_vp)],
    "wf_free": [_vp, _vp],
    "wf_ctx_trim": [_vp],
    "wf_memcpy_h2d": [_vp, _vp, _vp, ctypes.c_size_t],
    "wf_memcpy_d2h": [_vp, _vp, _vp, ctypes.c_size_t],
    "wf_memcpy_d2d": [_vp, _vp, _vp, ctypes.c_size_t],
    "wf_host_register": [_vp, _vp, ctypes.c_size_t],
    "wf_host_unregister": [_vp, _vp],
    "wf_fft_get_twiddles": [_vp, _int, _u32, _int, _vp],
    "wf_fft_evaluate_poly": [_vp, _int, _u32, _vp, _u32, _u32],
    "wf_fft_interpolate_poly": [_vp, _int, _u32, _vp, _u32, _u32],
    "wf_fft_evaluate_poly_with_offset": [_vp, _int, _u32, _vp, _u32, _vp, _u32, _vp],
    "wf_fft_interpolate_poly_with_offset": [_vp, _int, _u32, _vp, _u32, _vp],
    "wf_get_power_series_with_offset": [_vp, _int, _vp, _vp, _u64, _vp],
    "wf_batch_inversion": [_vp, _int, _vp, _u64, _vp],
    "wf_interpolate_columns": [_vp, _int, _u32, _vp, _u32, _u64, _u32],
    "wf_evaluate_polys_over": [_vp, _int, _u32, _vp, _u32, _u64, _u32, _u32, _vp, _vp],
    "wf_evaluate_columns_over": [_vp, _int, _u32, _vp, _u32, _u64, _u32, _u32, _vp, _vp, _u64],
    "wf_hash_columns": [_vp, _int, _int, _u32, _vp, _u32, _u64, _u64, _vp],
    "wf_hash_rows": [_vp, _int, _int, _u32, _vp, _u64, _u64, _u32, _u32, _u32, _vp],
    "wf_merkle_build": [_vp, _int, _vp, _u64, _vp],
    "wf_hash_merge_batch": [_vp, _int, _vp, _u64, _vp],
    "wf_hash_elements_batch": [_vp, _int, _int, _vp, _u64, _u64, _u32, _vp],
    "wf_hash_bytes_batch": [_vp, _int, _vp, _u64, _u64, _u64, _vp],
    "wf_hash_merge_many_batch": [_vp, _int, _vp, _u64, _u32, _vp],
    "wf_hash_merge_with_int_batch": [_vp, _int, _vp, _u64, _u64, _vp],
    "wf_grind": [_vp, _int, _vp, _u32, _u64, _u64, ctypes.POINTER(ctypes.c_uint64)],
    "wf_build_trace_commitment": [_vp, _int, _int, _u32, _vp, _u32, _u64, _u32, _u32, _vp, _u32, _u32, _int, _vp, _vp,
                                  _vp, _vp],
    "wf_rows_fetch": [_vp, _vp, _u64, _u32, _u32, _vp, _u32, _vp],
    "wf_evaluate_constraints": [_vp, _int, _int, _u32, _vp, _u64, _u32, _u32, _u32, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp],
    "wf_evaluate_constraints_assertions": [_vp, _int, _int, _u32, _vp, _u64, _u32, _u32, _u32, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "wf_evaluate_constraints_aux": [_vp, _int, _int, _u32, _vp, _u64, _vp, _u64, _u32, _u32, _u32, _vp, _vp, _u32, _vp, _vp, _vp, _vp,
                                    _u32, _vp, _vp, _vp, _vp, _vp, _vp],
    "wf_polys_evaluate_at": [_vp, _int, _u32, _u32, _vp, _u32, _u64, _u32, _vp, _u32, _vp],
    "wf_deep_compose": [_vp, _int, _u32, _vp, _u32, _u64, _vp, _u32, _u64, _vp, _u32, _u64, _u32, _vp, _vp, _vp, _vp],
    "wf_fri_layer_commit": [_vp, _int, _int, _u32, _vp, _u32, _u32, _vp, _vp, _vp, _vp],
    "wf_fri_apply_drp": [_vp, _int, _u32, _vp, _u32, _u32, _vp, _vp, _vp],
    "wf_fri_apply_drp_rows": [_vp, _int, _u32, _vp, _u32, _u32, _u64, _u64, _vp, _vp, _vp],
    "wf_fri_build_layers": [_vp, _int, _int, _u32, _vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _u32, _vp],
    "wf_coin_init": [_vp, _vp, _vp],
    "wf_coin_reseed": [_vp, _int, _vp, _vp, _vp],
    "wf_coin_draw": [_vp, _int, _int, _u32, _vp, _u32, _vp],
    "wf_coin_reseed_draw": [_vp, _int, _int, _u32, _vp, _vp, _vp, _vp],
    "wf_coin_read": [_vp, _vp, _vp, ctypes.POINTER(ctypes.c_uint64)],
    "wf_comm_get_unique_id": [_vp],
    "wf_comm_init_rank": [_vp, _vp, _int, _int, _vp],
    "wf_comm_init_loopback": [_vp, _int, _vp],
    "wf_comm_destroy": [_vp],
    "wf_comm_rank": [_vp],
    "wf_comm_size": [_vp],
    "wf_comm_all_gather": [_vp, _vp, _vp, _u64],
    "wf_comm_all_to_all": [_vp, _vp, _vp, _u64],
    "wf_comm_sharded_commit": [_vp, _int, _int, _u32, _vp, _u32, _u64, _u32, _u32, _vp, _int, _vp, _vp, _vp, _vp, _vp],
    "wf_comm_sharded_fri_layers": [_vp, _int, _int, _u32, _vp, _u32, _u32, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "wf_fri_apply_drp_rows_dev": [_vp, _int, _u32, _vp, _u32, _u32, _u64, _u64, _vp, _vp, _vp],
    "wf_coin_grind": [_vp, _int, _vp, _u32, _u32, _vp],
    "wf_coin_draw_integers": [_vp, _int, _vp, _vp, _u32, _u32, _vp],
    "wf_evaluate_constraints_dev": [_vp, _int, _int, _u32, _vp, _u64, _u32, _u32, _u32, _vp, _vp, _u32, _vp, _vp, _vp, _vp, _vp, _vp, _vp],
    "wf_polys_evaluate_at_dev": [_vp, _int, _u32, _u32, _vp, _u32, _u64, _u32, _vp, _int, _vp],
    "wf_deep_compose_dev": [_vp, _int, _u32, _vp, _u32, _u64, _vp, _u32, _u64, _vp, _u32, _u64, _u32, _vp, _vp, _vp],
}

_lib = None
_lock = threading.Lock()


class WfError(RuntimeError):
    """Non-zero status from the C ABI (the Rust shim would panic!/Err here)."""

    def __init__(self, status, where=""):
        self.status = status
        msg = load_library().wf_strerror(status).decode()
        super().__init__("%s: %s (status %d)" % (where or "winterfell_hip", msg, status))


def load_library():
    """Load libwinterfell_hip.so; raises if the HIP extension has not been built (no fallback)."""
    global _lib
    with _lock:
        if _lib is None:
            # torch must be imported first: its bundled libamdhip64.so carries the soname libamdhip64.so.7, so our
            # DT_NEEDED resolves to the runtime torch already loaded and the process has ONE HIP runtime (streams
            # and allocations are shared with torch).  Loaded the other way round the process gets two runtimes.
            import torch  # noqa: F401
            if not os.path.exists(LIB_PATH):
                raise RuntimeError(
                    "libwinterfell_hip.so is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                    "or `make -C winterfell_amd/csrc` — there is no CPU fallback")
            lib = ctypes.CDLL(LIB_PATH)
            for name, args in _PROTOS.items():
                fn = getattr(lib, name)
                fn.argtypes = args
                fn.restype = _int
            lib.wf_strerror.argtypes = [_int]
            lib.wf_strerror.restype = ctypes.c_char_p
            lib.wf_version.restype = _int
            lib.wf_row_width.argtypes = [_u32, _u32]
            lib.wf_row_width.restype = _u64
            lib.wf_last_device_status.restype = _u32
            _lib = lib
    return _lib


def _check(status, where):
    if status != 0:
        raise WfError(status, where)


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("winterfell_amd needs a HIP device (torch.cuda.is_available() is False); no CPU fallback")
    return torch


class Context:
    """Owns a wf_ctx bound to one GPU; kernels run on torch's current stream of that device."""

    def __init__(self, device=0):
        self.lib = load_library()
        torch = _torch()
        self.device = torch.device("cuda", device)
        h = _vp()
        s = torch.cuda.current_stream(self.device).cuda_stream          # born on torch's current stream: no private stream to throw away
        _check(self.lib.wf_ctx_create_on_stream(device, _vp(s), ctypes.byref(h)), "wf_ctx_create_on_stream")
        self.handle = h
        self._bound_stream = s
        _live.append(self)

    def use_torch_stream(self):
        """Bind the library's launches to torch's CURRENT stream of this device.  Re-checked on every call(): torch work
        issued around the library calls (empty / clone / reductions / copies) runs on whatever stream is current, e.g.
        inside `with torch.cuda.stream(s)`, and must stay ordered with the kernels."""
        torch = _torch()
        s = torch.cuda.current_stream(self.device).cuda_stream
        if s != getattr(self, "_bound_stream", None):
            _check(self.lib.wf_ctx_set_stream(self.handle, _vp(s)), "wf_ctx_set_stream")
            self._bound_stream = s

    def sync(self):
        _check(self.lib.wf_ctx_sync(self.handle), "wf_ctx_sync")

    def close(self):
        if self.handle is not None:
            self.lib.wf_ctx_destroy(self.handle)
            self.handle = None
        if self in _live:
            _live.remove(self)

    # ---- buffers --------------------------------------------------------------------------------------
    def to_device(self, arr):
        """numpy uint64/uint8 array (or torch tensor) -> torch tensor in HBM (int64 / uint8 bit containers).  Host data crosses
        through wf_memcpy_h2d — the library's page-locked bounce buffers — never through a runtime copy out of PAGEABLE memory:
        ROCclr pins such a range in place and finds the pinned object again by address alone after the buffer was freed and
        re-allocated (a GPU page fault at a host address, see csrc/context.hip and DESIGN.md section 9)."""
        torch = _torch()
        if isinstance(arr, torch.Tensor):
            if arr.device.type == "cuda":
                return arr.to(self.device).contiguous()
            arr = arr.numpy()
        a = np.ascontiguousarray(arr)
        if a.dtype == np.uint64:
            a = a.view(np.int64)
        elif a.dtype != np.uint8 and a.dtype != np.int64:
            raise TypeError("expected uint64 / uint8 data, got %s" % a.dtype)
        out = torch.empty(a.shape, dtype=torch.int64 if a.dtype == np.int64 else torch.uint8, device=self.device)
        if a.nbytes:
            self.call("wf_memcpy_h2d", _vp(out.data_ptr()), _vp(a.ctypes.data), a.nbytes)
        return out

    def empty_u64(self, *shape):
        return _torch().empty(shape, dtype=_torch().int64, device=self.device)

    def empty_u8(self, *shape):
        return _torch().empty(shape, dtype=_torch().uint8, device=self.device)

    def to_host(self, t):
        """device tensor -> numpy array (uint64 view of the int64 bit container), through wf_memcpy_d2h (see to_device)."""
        torch = _torch()
        t = t.detach()
        if t.device.type != "cuda":
            a = t.numpy()
            return a.view(np.uint64) if a.dtype == np.int64 else a
        if t.dtype not in (torch.int64, torch.uint8):
            raise TypeError("expected an int64 / uint8 tensor, got %s" % t.dtype)
        t = t.contiguous()
        a = np.empty(tuple(t.shape), dtype=np.int64 if t.dtype == torch.int64 else np.uint8)
        if a.nbytes:
            self.call("wf_memcpy_d2h", _vp(a.ctypes.data), _vp(t.data_ptr()), a.nbytes)
        return a.view(np.uint64) if a.dtype == np.int64 else a

    def shader_clock_mhz(self, spin_us=500):
        """the shader clock while the work already queued on this context's stream runs (wf_debug_shader_clock)"""
        mhz = ctypes.c_double(0.0)
        self.call("wf_debug_shader_clock", spin_us, ctypes.byref(mhz))
        return float(mhz.value)

    def prof_enable(self, on=True):
        """True / 1: every launch bracketed by events; 2: one event pair around all the launches until prof_collect ("__span__")"""
        _check(self.lib.wf_prof_enable(self.handle, int(on)), "wf_prof_enable")

    def prof_collect(self):
        """-> {kernel_name: (launches, total_ms)} measured with HIP events on the context's stream."""
        buf = ctypes.create_string_buffer(1 << 16)
        _check(self.lib.wf_prof_collect(self.handle, buf, len(buf)), "wf_prof_collect")
        out = {}
        for line in buf.value.decode().splitlines():
            name, cnt, ms = line.split()
            out[name] = (int(cnt), float(ms))
        return out

    def call(self, name, *args):
        self.use_torch_stream()
        _check(getattr(self.lib, name)(self.handle, *args), name)


def ptr(t):
    return _vp(t.data_ptr())


def torch_u64():
    """the dtype the library's 64-bit words are held in on the torch side (the int64 bit container Context.empty_u64 allocates)"""
    return _torch().int64


_default = {}
_live = []


@atexit.register
def _shutdown():
    # destroy contexts while the HIP runtime (and torch's streams) are still alive
    for c in list(_live):
        try:
            c.close()
        except Exception:
            pass
    _default.clear()


def default_context(device=None):
    torch = _torch()
    if device is None:
        device = torch.cuda.current_device()
    if device not in _default:
        _default[device] = Context(device)
    return _default[device]
