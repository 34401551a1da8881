"""ctypes binding of ``libwwhip.so`` (the C ABI declared in ``include/wwhip.h``).

There is no CPU fallback: if the shared object is missing, or no gfx950 device is visible
when a context is requested, the call raises.  Status codes map to the exception types the
reference's callers see from TFLite/NumPy (``ValueError`` for bad shapes and arguments,
``RuntimeError`` for runtime failures).
"""
from __future__ import annotations

import ctypes as C
import json
import os
import threading
from typing import Dict, Optional

import numpy as np

_PKG = os.path.dirname(os.path.abspath(__file__))
# WWHIP_LIB: development only - an alternative build of the same library (kernel A/B comparisons)
LIB_PATH = os.environ.get("WWHIP_LIB") or os.path.join(_PKG, "libwwhip.so")

WW_OK, WW_EINVAL, WW_EBLOB, WW_EHIP, WW_ENOMEM, WW_ESTATE, WW_ENODEVICE, WW_EINTERNAL = 0, -1, -2, -3, -4, -5, -6, -7
KIND_CRNN, KIND_WAVENET = 1, 2
PRECISION_FP32, PRECISION_BF16X3 = 0, 1
OPT_CRNN_SPLIT_AT, OPT_CRNN_SLIDE_MIN, OPT_CRNN_TAIL_MFMA, OPT_WAVENET_ROWMAJOR = 1, 2, 3, 4
STREAM_FULL_RECOMPUTE, STREAM_TWO_LAUNCH, STREAM_SYNC_WAIT = 1, 2, 4
ABI = 4  # include/wwhip.h: WW_ABI - the signatures this binding was written against


class ModelInfo(C.Structure):
    _fields_ = [(n, C.c_int32) for n in
                ("kind", "window", "n_mel", "n_bins", "n_out", "enc_rows", "enc_width", "reserved")]


class PipelineState(C.Structure):
    """``ww_pipeline_state`` (include/wwhip.h): the arrays of the three banked stages, for ``ww_pipeline_bank_step``."""
    _fields_ = ([(n, C.c_void_p) for n in ("is_speech", "is_active", "raw", "run_value", "run_length", "wake_was_speech", "posterior_max", "post",
                                           "n_post", "timeout_was_speech", "active_frames", "fired_ids", "fall_ids", "deact_ids")]
                + [(n, C.c_double) for n in ("threshold", "min_frames", "max_frames")]
                + [(n, C.c_int32) for n in ("rise_frames", "fall_frames", "n_vad_changed", "n_fired", "n_fall", "n_deact")])


class FrontendParams(C.Structure):
    _fields_ = [("pcm_divisor", C.c_float), ("clip", C.c_int32), ("pre_emphasis", C.c_float),
                ("hop", C.c_int32), ("precise", C.c_int32)]


# name -> (restype, argtypes); every symbol include/wwhip.h declares
_vp, _i32, _i64, _f32, _f64, _sz = C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_double, C.c_size_t
_P = C.POINTER
SYMBOLS: Dict[str, tuple] = {
    "ww_ctx_create": (C.c_int, [C.c_int, _vp, _P(_vp)]),
    "ww_ctx_destroy": (C.c_int, [_vp]),
    "ww_ctx_synchronize": (C.c_int, [_vp]),
    "ww_ctx_stream": (_vp, [_vp]),
    "ww_last_error": (C.c_char_p, [_vp]),
    "ww_version": (C.c_char_p, []),
    "ww_runtime_info": (C.c_int, [_P(_i32), _P(_i32), _P(_i32)]),
    "ww_host_stage_i16": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _i64, _i64, _i32]),
    "ww_uploader_create": (C.c_int, [_vp, _i32, _i32, _P(_vp)]),
    "ww_uploader_destroy": (C.c_int, [_vp]),
    "ww_uploader_submit": (C.c_int, [_vp, _i64, _i64, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _P(_i64)]),
    "ww_uploader_poll": (C.c_int, [_vp, _i64]),
    "ww_uploader_wait": (C.c_int, [_vp, _i64, _vp]),
    "ww_profile_enable": (C.c_int, [_vp, C.c_int]),
    "ww_profile_read": (C.c_int, [_vp, C.c_char_p, _sz]),
    "ww_timer_start": (C.c_int, [_vp]),
    "ww_timer_stop": (C.c_int, [_vp, _P(_f32)]),
    "ww_model_load": (C.c_int, [_vp, _vp, _sz, _P(_vp)]),
    "ww_model_free": (C.c_int, [_vp]),
    "ww_model_get_info": (C.c_int, [_vp, _P(ModelInfo)]),
    "ww_model_set_precision": (C.c_int, [_vp, C.c_int]),
    "ww_model_set_option": (C.c_int, [_vp, C.c_int, _i64]),
    "ww_num_frames": (_i64, [_i64, _i32]),
    "ww_logmel": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _P(FrontendParams), _vp, _vp]),
    "ww_logmel_f32": (C.c_int, [_vp, _vp, _vp, _vp, _i32, _P(FrontendParams), _vp, _vp]),
    "ww_stft_mag": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp]),
    "ww_filter_apply": (C.c_int, [_vp, _vp, _vp, _i64, _vp]),
    "ww_detect": (C.c_int, [_vp, _vp, _vp, _i32, _vp]),
    "ww_logmel_dev": (C.c_int, [_vp, _vp, _vp, _vp, _vp, _i32, _i64, _i64, _P(FrontendParams), _vp]),
    "ww_forward": (C.c_int, [_vp, _vp, _vp, _i32, _vp]),
    "ww_forward_enc": (C.c_int, [_vp, _vp, _vp, _i32, _vp, _vp]),
    "ww_slide_forward": (C.c_int, [_vp, _vp, _vp, _i64, _i32, _vp, _P(_i64)]),
    "ww_forward_windows_dev": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _i32, _vp]),
    "ww_forward_segments_dev": (C.c_int, [_vp, _vp, _vp, _i64, _vp, _vp, _i32, _i32, _vp]),
    "ww_clips_forward_dev": (C.c_int, [_vp, _vp, _vp, _i32, _i32, _P(FrontendParams), _vp]),
    "ww_stream_create": (C.c_int, [_vp, _vp, _i32, _P(FrontendParams), C.c_uint32, _P(_vp)]),
    "ww_stream_destroy": (C.c_int, [_vp]),
    "ww_stream_step": (C.c_int, [_vp, _vp, _vp, _vp, _vp]),
    "ww_stream_reset": (C.c_int, [_vp, _vp, _i32]),
    "ww_stream_timeline": (C.c_int, [_vp, _vp, _P(_i64), _i32]),
    "ww_vad_bank_step": (C.c_int, [_i32, _vp, _i32, _i32, _vp, _vp, _vp, _vp]),
    "ww_trigger_bank_step": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _f64, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ww_timeout_bank_step": (C.c_int, [_i32, _vp, _vp, _vp, _vp, _f64, _f64, _vp, _vp]),
    "ww_stream_step_trigger": (C.c_int, [_vp, _vp, _vp, _vp, _f64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    "ww_pipeline_bank_step": (C.c_int, [_vp, _vp, _P(PipelineState)]),
    "ww_superframe_smooth": (C.c_int, [_vp, _vp, _i64, _i32, _f32, _i32, _vp, _vp]),
    "ww_far_frr": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _i32, _f64, _f64, _vp, _vp, _vp, _vp]),
    "ww_far_frr_dev": (C.c_int, [_vp, _vp, _i64, _vp, _i64, _i32, _vp, _i32, _f64, _f64, _vp, _vp, _vp, _vp]),
    "ww_posterior_pick_dev": (C.c_int, [_vp, _vp, _i64, _i32, _i32, _vp, _i64, _vp]),
}

_lib: Optional[C.CDLL] = None
_lock = threading.Lock()

# Native objects must be released before the interpreter (and with it the HIP runtime) is torn
# down, and in dependency order: uploaders and stream banks, then models, then contexts.  Objects register
# here; an atexit hook closes them; after that close() / __del__ are no-ops.
import atexit
import weakref

_live = {"uploaders": weakref.WeakSet(), "streams": weakref.WeakSet(), "models": weakref.WeakSet(), "contexts": weakref.WeakSet()}
_shutdown = False


def register(kind: str, obj) -> None:
    _live[kind].add(obj)


def is_shutdown() -> bool:
    return _shutdown


def _close_all() -> None:
    global _shutdown
    for kind in ("uploaders", "streams", "models", "contexts"):
        for obj in list(_live[kind]):
            try:
                obj.close()
            except Exception:
                pass
    _shutdown = True


atexit.register(_close_all)


def _torch_first() -> None:
    """PyTorch-ROCm wheels carry their own HIP runtime.  It works beside the system one that ``libwwhip.so`` links
    (``/opt/rocm``) only when it is loaded FIRST; loaded second it reports "No HIP GPUs are available" (measured on
    this image: torch 2.10+rocm7.0 beside ROCm 7.2).  The library does not need torch - but the evaluators hold their
    device buffers in torch tensors and callers mix the two freely, so when torch is installed it goes first."""
    import importlib.util
    import sys
    if "torch" in sys.modules:
        return
    try:
        if importlib.util.find_spec("torch") is not None:
            import torch  # noqa: F401
    except Exception:  # a broken torch install must not take the library down with it
        pass


def load() -> C.CDLL:
    """Load ``libwwhip.so``; raises ``RuntimeError`` if it has not been built."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not os.path.isfile(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950). "
                "There is no CPU fallback for the wake-word hot path.")
        _torch_first()
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)  # AttributeError if the symbol is not exported
            fn.restype = res
            fn.argtypes = args
        _check_abi(lib)
        _lib = lib
        _check_runtime(lib)
        return lib


def _check_abi(lib: C.CDLL) -> None:
    """``ww_version()`` carries "ABI <n>": a library of another ABI (a stale ``WWHIP_LIB`` variant, say) has entry points with
    other signatures behind the same names - refuse it here instead of crashing inside the first call that differs."""
    import re
    text = (lib.ww_version() or b"").decode("utf-8", "replace")
    m = re.search(r"ABI (\d+)", text)
    if m is None or int(m.group(1)) != ABI:
        raise RuntimeError(f"{LIB_PATH} reports {text!r}; this binding needs ABI {ABI} (include/wwhip.h: WW_ABI). Rebuild it "
                           "with `python __graft_entry__.py`.")


def runtime_info() -> Dict[str, int]:
    """HIP version the library was built against and the one it is running on (``ww_runtime_info``)."""
    b, r, d = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    load().ww_runtime_info(C.byref(b), C.byref(r), C.byref(d))
    return {"built": int(b.value), "runtime": int(r.value), "driver": int(d.value)}


def _check_runtime(lib: C.CDLL) -> None:
    """The library's libamdhip64 binds to whichever copy the process loaded first (torch's bundled runtime when torch
    is imported, see :func:`_torch_first`).  A different MAJOR version than the build headers means different struct
    layouts behind the same symbols: say so loudly instead of failing somewhere inside a launch."""
    import warnings
    b, r, d = C.c_int32(0), C.c_int32(0), C.c_int32(0)
    rc = lib.ww_runtime_info(C.byref(b), C.byref(r), C.byref(d))
    if rc != WW_OK or r.value == 0:
        return  # no runtime answer (e.g. no GPU in a build container): Context() will fail loudly if it matters
    if b.value // 10_000_000 != r.value // 10_000_000:
        warnings.warn(f"libwwhip.so was built against HIP {b.value} but runs on HIP runtime {r.value} "
                      f"(driver {d.value}): major versions differ", RuntimeWarning, stacklevel=3)


def ptr(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def addr(a: np.ndarray) -> C.c_void_p:
    """The address of an array that stays where it is (a bank's state, a ContextBank's flags), taken ONCE: the per-tick stage
    calls pass these objects as they are (a tick is host-paced; ``a.ctypes.data_as`` per argument and call is microseconds)."""
    if not a.flags.c_contiguous:
        raise ValueError("a contiguous array is needed")
    return C.c_void_p(a.ctypes.data)


def raise_for(rc: int, ctx_handle) -> None:
    if rc == WW_OK:
        return
    msg = load().ww_last_error(ctx_handle)
    text = msg.decode("utf-8", "replace") if msg else f"libwwhip error {rc}"
    if rc in (WW_EINVAL, WW_EBLOB):
        raise ValueError(text)
    if rc == WW_ENOMEM:
        raise MemoryError(text)
    raise RuntimeError(text)


class Context:
    """One ``ww_ctx``: a HIP stream + workspace on one gfx950 device."""

    def __init__(self, device: int = 0, stream: Optional[int] = None) -> None:
        lib = load()
        h = C.c_void_p()
        rc = lib.ww_ctx_create(device, C.c_void_p(stream) if stream else None, C.byref(h))
        if rc != WW_OK:
            raise_for(rc, None)
        self._h = h
        self.device = device
        register("contexts", self)

    @property
    def handle(self):
        return self._h

    def synchronize(self) -> None:
        raise_for(load().ww_ctx_synchronize(self._h), self._h)

    @property
    def stream(self) -> int:
        return int(load().ww_ctx_stream(self._h) or 0)

    def profile(self, on: bool) -> None:
        raise_for(load().ww_profile_enable(self._h, int(on)), self._h)

    def profile_read(self) -> dict:
        buf = C.create_string_buffer(1 << 16)
        raise_for(load().ww_profile_read(self._h, buf, len(buf)), self._h)
        return json.loads(buf.value.decode())

    def timer_start(self) -> None:
        raise_for(load().ww_timer_start(self._h), self._h)

    def timer_stop(self) -> float:
        ms = C.c_float()
        raise_for(load().ww_timer_stop(self._h, C.byref(ms)), self._h)
        return float(ms.value)

    def close(self) -> None:
        if self._h and not _shutdown:
            load().ww_ctx_destroy(self._h)
        self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


class Uploader:
    """One ``ww_uploader``: a library thread that assembles chunks of int16 samples in page-locked slots and sends them to the
    context's device on a copy stream of its own (``include/wwhip.h``)."""

    def __init__(self, ctx: Context, slots: int = 3, copy_threads: int = 8) -> None:
        h = C.c_void_p()
        raise_for(load().ww_uploader_create(ctx.handle, int(slots), int(copy_threads), C.byref(h)), ctx.handle)
        self._h, self.device = h, ctx.device
        register("uploaders", self)

    def submit(self, total: int, dst_off: np.ndarray, src: np.ndarray, count: np.ndarray, d_pcm_ptr: int, meta: np.ndarray,
               d_meta_ptr: int) -> int:
        """Queue one chunk (arrays int64, contiguous; ``src`` = addresses of the runs' first samples); returns the ticket."""
        t = C.c_int64(0)
        for a in (dst_off, src, count, meta):
            if a.dtype != np.int64 or not a.flags.c_contiguous:
                raise ValueError("Uploader.submit takes contiguous int64 arrays")
        if not (len(dst_off) == len(src) == len(count)):
            raise ValueError("dst_off, src and count must have one entry per run")
        rc = load().ww_uploader_submit(self._h, int(total), len(dst_off), ptr(dst_off), ptr(src), ptr(count), C.c_void_p(d_pcm_ptr),
                                       len(meta), ptr(meta), C.c_void_p(d_meta_ptr), C.byref(t))
        if rc != WW_OK:
            raise ValueError(f"ww_uploader_submit refused the chunk ({rc})")
        return int(t.value)

    def done(self, ticket: int) -> bool:
        return load().ww_uploader_poll(self._h, int(ticket)) == 1

    def wait(self, ticket: int, ctx: Context) -> None:
        """Block until the ticket's copies are enqueued; ``ctx``'s stream then waits for them on the device."""
        raise_for(load().ww_uploader_wait(self._h, int(ticket), ctx.handle), ctx.handle)

    def close(self) -> None:
        if self._h and not _shutdown:
            load().ww_uploader_destroy(self._h)
        self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass


_tls = threading.local()  # per host thread: {"ctx": {device: Context}}; dropped - and with it the contexts - when the thread ends


def default_context(device: int = 0) -> Context:
    """The calling thread's context on ``device`` for the drop-in classes: include/wwhip.h asks for one ``ww_ctx`` per
    host thread (the reference itself is single-threaded: SURVEY 8b 'Threading / ownership'), so two threads that use
    the drop-in classes never share a stream or a workspace.  The context lives in thread-local storage: when a short-lived
    thread ends, its context (stream, events, arenas) is released with it instead of staying on the GPU until the process
    exits."""
    per_thread = _tls.__dict__.setdefault("ctx", {})
    ctx = per_thread.get(device)
    if ctx is None or ctx.handle is None:
        ctx = Context(device)
        per_thread[device] = ctx
    return ctx
