"""Host-side handle on one uploaded model: thin, typed wrappers over the C ABI.

Everything here forwards to ``libwwhip.so``; nothing is computed in Python/NumPy apart from
argument marshalling.  The reference-shaped classes (``Filter``, ``TFLiteModel``,
``WakewordTrigger``, ``get_posterior`` ...) are built on top of this in the sibling modules.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from . import weights as W


def frontend_params(pcm_divisor: float = 32767.0, clip: bool = True, pre_emphasis: float = 0.0,
                    hop: int = 160, precise: bool = True) -> _lib.FrontendParams:
    return _lib.FrontendParams(float(pcm_divisor), int(bool(clip)), float(pre_emphasis), int(hop), int(bool(precise)))


class Engine:
    """A model directory (filter/encode/detect ``.tflite``) resident on one MI355X."""

    def __init__(self, model_dir: str, device: int = 0, ctx: Optional[_lib.Context] = None,
                 precision: str = "fp32", weights_fp16: bool = False) -> None:
        self._weights_fp16 = bool(weights_fp16)
        self.bundle = W.load_model_dir(model_dir)
        if weights_fp16:  # the reference's float16-quantised model variant (weights.quantize_fp16)
            self.bundle = W.quantize_fp16(self.bundle)
        self.blob = W.pack_blob(self.bundle)
        self.ctx = ctx if ctx is not None else _lib.default_context(device)
        self._lib = _lib.load()
        h = C.c_void_p()
        buf = np.frombuffer(self.blob, dtype=np.uint8)
        _lib.raise_for(self._lib.ww_model_load(self.ctx.handle, _lib.ptr(buf), buf.size, C.byref(h)), self.ctx.handle)
        self._model = h
        info = _lib.ModelInfo()
        _lib.raise_for(self._lib.ww_model_get_info(h, C.byref(info)), self.ctx.handle)
        self.kind = info.kind
        self.window = info.window
        self.n_mel = info.n_mel
        self.n_bins = info.n_bins
        self.n_out = info.n_out
        self.enc_shape = (info.enc_rows, info.enc_width)
        self.model_dir = model_dir
        _lib.register("models", self)
        self._options = {"crnn_split_at": 1024, "crnn_slide_min": 64, "crnn_tail_mfma": 1, "wavenet_rowmajor": 0}  # the library's defaults
        self.precision = "fp32"
        if precision != "fp32":
            self.set_precision(precision)

    def set_precision(self, precision: str) -> None:
        """``"fp32"`` (default, fp32 MFMA) or ``"bf16x3"`` (Wavenet blocks as three bf16 MFMAs on
        split operands, fp32 accumulate; see ``ww_model_set_precision`` in include/wwhip.h)."""
        modes = {"fp32": _lib.PRECISION_FP32, "bf16x3": _lib.PRECISION_BF16X3}
        if precision not in modes:
            raise ValueError(f"precision must be one of {sorted(modes)}")
        self._chk(self._lib.ww_model_set_precision(self._model, modes[precision]))
        self.precision = precision

    def set_option(self, key: str, value: int) -> None:
        """Per-model dispatch options (``ww_model_set_option``): ``"crnn_split_at"`` - explicit-window launches above
        this many windows take front + tail kernels (0 = always one fused kernel); ``"crnn_slide_min"`` - regular
        sliding windows take the once-per-sequence form from this many windows on (0 = never); ``"crnn_tail_mfma"`` - the recurrences
        of those two forms for sixteen windows per workgroup on the matrix pipe: 1 (default) from 9,216 windows per launch on,
        2 always, 0 never (one window per workgroup on the vector ALU); ``"wavenet_rowmajor"`` - 1: the fp32 Wavenet's row-major block loop of rounds 1-2 instead of the
        transposed one."""
        keys = {"crnn_split_at": _lib.OPT_CRNN_SPLIT_AT, "crnn_slide_min": _lib.OPT_CRNN_SLIDE_MIN,
                "crnn_tail_mfma": _lib.OPT_CRNN_TAIL_MFMA, "wavenet_rowmajor": _lib.OPT_WAVENET_ROWMAJOR}
        if key not in keys:
            raise ValueError(f"option must be one of {sorted(keys)}")
        self._chk(self._lib.ww_model_set_option(self._model, keys[key], int(value)))
        self._options[key] = int(value)

    def options(self, **kv):
        """Context manager: the given options for the duration of a ``with`` block, the previous values afterwards."""
        import contextlib

        @contextlib.contextmanager
        def scope():
            old = {k: self._options[k] for k in kv}
            try:
                for k, v in kv.items():
                    self.set_option(k, v)
                yield self
            finally:
                for k, v in old.items():
                    self.set_option(k, v)
        return scope()

    def lane(self, k: int) -> "Engine":
        """Lane ``k`` of this engine: ``self`` for 0, otherwise a twin on a context (HIP stream) of its own - the same files, the
        same precision and dispatch options (brought in line at every call), its own 0.9 MB of weights.  Independent launches dealt
        to the lanes run beside each other on the GPU, which one stream's order forbids: the sharded evaluators give consecutive
        chunks to alternating lanes (wwhip/evaluate.py, as bench.py does with whole steps)."""
        if k == 0:
            return self
        lanes = self.__dict__.setdefault("_lanes", {})
        e = lanes.get(k)
        if e is None or e.handle is None:
            e = lanes[k] = Engine(self.model_dir, device=self.ctx.device, ctx=_lib.Context(self.ctx.device), precision=self.precision,
                                  weights_fp16=self._weights_fp16)
        if e.precision != self.precision:
            e.set_precision(self.precision)
        for key, v in self._options.items():
            if e._options[key] != v:
                e.set_option(key, v)
        return e

    # ------------------------------------------------------------------ properties
    @property
    def handle(self):
        return self._model

    @property
    def posterior_index(self) -> int:
        return self.bundle.posterior_index

    @property
    def is_crnn(self) -> bool:
        return self.kind == _lib.KIND_CRNN

    def close(self) -> None:
        for e in self.__dict__.pop("_lanes", {}).values():  # the twins on contexts of their own (lane) go with their engine
            e.close()
        if self._model and not _lib.is_shutdown():
            self._lib.ww_model_free(self._model)
        self._model = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _chk(self, rc: int) -> None:
        _lib.raise_for(rc, self.ctx.handle)

    # ------------------------------------------------------------------ front end
    def num_frames(self, n_samples: int, hop: int = 160) -> int:
        return int(self._lib.ww_num_frames(int(n_samples), int(hop)))

    def logmel(self, pcm: Sequence[np.ndarray], fp: Optional[_lib.FrontendParams] = None) -> List[np.ndarray]:
        """List of int16 (or float32) utterances -> list of ``[frames, 40]`` log-mel arrays."""
        fp = fp or frontend_params()
        if len(pcm) == 0:
            return []
        is_f32 = np.asarray(pcm[0]).dtype != np.int16
        dt = np.float32 if is_f32 else np.int16
        arrs = [np.ascontiguousarray(p, dtype=dt).ravel() for p in pcm]
        # pad every utterance start to a multiple of 8 samples so device loads stay aligned
        offs = np.zeros(len(arrs) + 1, np.int64)
        for i, a in enumerate(arrs):
            offs[i + 1] = offs[i] + a.size
        flat = np.concatenate(arrs) if arrs else np.zeros(0, dt)
        foffs = np.zeros(len(arrs) + 1, np.int64)
        total = sum(self.num_frames(a.size, fp.hop) for a in arrs)
        mel = np.empty((total, self.n_mel), np.float32)
        fn = self._lib.ww_logmel_f32 if is_f32 else self._lib.ww_logmel
        self._chk(fn(self.ctx.handle, self._model, _lib.ptr(flat), _lib.ptr(offs), len(arrs), C.byref(fp),
                     _lib.ptr(mel), _lib.ptr(foffs)))
        return [mel[foffs[i]:foffs[i + 1]] for i in range(len(arrs))]

    def stft_mag(self, frames: np.ndarray, precise: bool = True) -> np.ndarray:
        f = np.ascontiguousarray(frames, dtype=np.float32).reshape(-1, 512)
        mag = np.empty((f.shape[0], self.n_bins), np.float32)
        self._chk(self._lib.ww_stft_mag(self.ctx.handle, self._model, _lib.ptr(f), f.shape[0], int(precise), _lib.ptr(mag)))
        return mag

    def filter_apply(self, mag: np.ndarray) -> np.ndarray:
        """``[n, 257]`` STFT magnitudes -> ``[n, 40]`` log-mel (filter.tflite alone)."""
        a = np.ascontiguousarray(mag, dtype=np.float32)
        if a.ndim != 2 or a.shape[1] != self.n_bins:
            raise ValueError(f"Cannot set tensor: Dimension mismatch. Got {a.shape} but expected (n, {self.n_bins})")
        mel = np.empty((a.shape[0], self.n_mel), np.float32)
        self._chk(self._lib.ww_filter_apply(self.ctx.handle, self._model, _lib.ptr(a), a.shape[0], _lib.ptr(mel)))
        return mel

    def detect(self, enc: np.ndarray) -> np.ndarray:
        """Encoder outputs ``[n, enc_rows, enc_width]`` -> detect rows ``[n, n_out]`` (detect.tflite alone)."""
        e = np.ascontiguousarray(enc, dtype=np.float32)
        per = self.enc_shape[0] * self.enc_shape[1]
        if e.size % per != 0 or e.size == 0:
            raise ValueError(f"Cannot set tensor: Dimension mismatch. Got {e.shape} but expected (n,) + {self.enc_shape}")
        n = e.size // per
        out = np.empty((n, self.n_out), np.float32)
        self._chk(self._lib.ww_detect(self.ctx.handle, self._model, _lib.ptr(e), n, _lib.ptr(out)))
        return out

    # ------------------------------------------------------------------ models
    def forward(self, windows: np.ndarray, want_enc: bool = False):
        """``[B, window, 40]`` -> detect rows ``[B, n_out]`` (and encoder output)."""
        w = np.ascontiguousarray(windows, dtype=np.float32)
        if w.ndim == 2:
            w = w[None]
        if w.ndim != 3 or w.shape[1] != self.window or w.shape[2] != self.n_mel:
            raise ValueError(f"Cannot set tensor: Dimension mismatch. Got {tuple(w.shape[1:])} but expected "
                             f"{(self.window, self.n_mel)} per window")
        out = np.empty((w.shape[0], self.n_out), np.float32)
        enc = np.empty((w.shape[0],) + self.enc_shape, np.float32) if want_enc else None
        self._chk(self._lib.ww_forward_enc(self.ctx.handle, self._model, _lib.ptr(w), w.shape[0], _lib.ptr(out), _lib.ptr(enc)))
        return (out, enc) if want_enc else out

    def slide_forward(self, mel: np.ndarray, hop: int = 2) -> np.ndarray:
        m = np.ascontiguousarray(mel, dtype=np.float32)
        if m.ndim != 2 or m.shape[1] != self.n_mel:
            raise ValueError(f"mel must be [rows, {self.n_mel}]")
        rows = m.shape[0]
        nw = (rows - self.window) // hop + 1 if rows >= self.window else 0
        out = np.empty((nw, self.n_out), np.float32)
        got = C.c_int64(0)
        self._chk(self._lib.ww_slide_forward(self.ctx.handle, self._model, _lib.ptr(m), rows, int(hop), _lib.ptr(out), C.byref(got)))
        assert got.value == nw
        return out

    # ------------------------------------------------------------------ evaluator
    def far_frr(self, pos: np.ndarray, neg: np.ndarray, thresholds: np.ndarray, num_wakewords: float, hours: float,
                window: int = 30, want_smoothed: bool = False):
        pos = np.ascontiguousarray(pos, dtype=np.float32).ravel()
        neg = np.ascontiguousarray(neg, dtype=np.float32).ravel()
        thr = np.ascontiguousarray(thresholds, dtype=np.float64).ravel()
        frr = np.empty(thr.size, np.float64)
        fa = np.empty(thr.size, np.float64)
        cnt = np.empty(thr.size, np.int64)
        sm = np.empty(neg.size, np.float64) if want_smoothed else None
        self._chk(self._lib.ww_far_frr(self.ctx.handle, _lib.ptr(pos), pos.size, _lib.ptr(neg), neg.size, int(window),
                                       _lib.ptr(thr), thr.size, float(num_wakewords), float(hours), _lib.ptr(frr),
                                       _lib.ptr(fa), _lib.ptr(cnt), _lib.ptr(sm)))
        return (frr, fa, cnt, sm) if want_smoothed else (frr, fa, cnt)

    def far_frr_dev(self, d_pos_ptr: int, n_pos: int, d_neg_ptr: int, n_neg: int, thresholds: np.ndarray, num_wakewords: float,
                    hours: float, window: int = 30):
        """:meth:`far_frr` over posteriors that are already on the device (``ww_far_frr_dev``): only the counters come back."""
        thr = np.ascontiguousarray(thresholds, dtype=np.float64).ravel()
        frr, fa, cnt = np.empty(thr.size, np.float64), np.empty(thr.size, np.float64), np.empty(thr.size, np.int64)
        self._chk(self._lib.ww_far_frr_dev(self.ctx.handle, C.c_void_p(d_pos_ptr), int(n_pos), C.c_void_p(d_neg_ptr), int(n_neg),
                                           int(window), _lib.ptr(thr), thr.size, float(num_wakewords), float(hours), _lib.ptr(frr),
                                           _lib.ptr(fa), _lib.ptr(cnt), None))
        return frr, fa, cnt

    def posterior_pick_dev(self, d_rows_ptr: int, n: int, d_out_ptr: int, d_seg_offs_ptr: int = 0, n_seg: int = 0) -> None:
        """Element ``posterior_index`` of ``n`` detect rows on the device -> ``d_out``: one value per row, or - with a device
        table of ``n_seg + 1`` row offsets - the maximum of each run (``ww_posterior_pick_dev``).  Enqueued, not waited for."""
        self._chk(self._lib.ww_posterior_pick_dev(self.ctx.handle, C.c_void_p(d_rows_ptr), int(n), self.n_out, self.posterior_index,
                                                  C.c_void_p(d_seg_offs_ptr) if d_seg_offs_ptr else None, int(n_seg),
                                                  C.c_void_p(d_out_ptr)))

    # ------------------------------------------------------------------ device-resident paths (torch plumbing)
    def clips_forward_dev(self, d_pcm_ptr: int, n_clips: int, samples_per_clip: int, d_out_ptr: int,
                          fp: Optional[_lib.FrontendParams] = None) -> None:
        fp = fp or frontend_params()
        self._chk(self._lib.ww_clips_forward_dev(self.ctx.handle, self._model, C.c_void_p(d_pcm_ptr), int(n_clips),
                                                 int(samples_per_clip), C.byref(fp), C.c_void_p(d_out_ptr)))

    def logmel_dev(self, d_pcm_ptr: int, d_sample_offs_ptr: int, d_frame_offs_ptr: int, n_utt: int, total_frames: int,
                   max_frames: int, d_mel_ptr: int, fp: Optional[_lib.FrontendParams] = None) -> None:
        fp = fp or frontend_params()
        self._chk(self._lib.ww_logmel_dev(self.ctx.handle, self._model, C.c_void_p(d_pcm_ptr), C.c_void_p(d_sample_offs_ptr),
                                          C.c_void_p(d_frame_offs_ptr), int(n_utt), int(total_frames), int(max_frames),
                                          C.byref(fp), C.c_void_p(d_mel_ptr)))

    def forward_windows_dev(self, d_mel_ptr: int, mel_rows: int, d_win_row_ptr: int, d_win_valid_ptr: int,
                            n_windows: int, d_out_ptr: int) -> None:
        self._chk(self._lib.ww_forward_windows_dev(self.ctx.handle, self._model, C.c_void_p(d_mel_ptr), int(mel_rows),
                                                   C.c_void_p(d_win_row_ptr), C.c_void_p(d_win_valid_ptr), int(n_windows),
                                                   C.c_void_p(d_out_ptr)))

    def forward_segments_dev(self, d_mel_ptr: int, mel_rows: int, seg_row0: np.ndarray, seg_nw: np.ndarray, hop: int,
                             d_out_ptr: int) -> None:
        """Several mel sequences in one device buffer, each slid over with ``hop`` (``ww_forward_segments_dev``): sequence
        ``s`` has ``seg_nw[s]`` complete windows starting at row ``seg_row0[s]``; detect rows go to ``d_out`` sequence by
        sequence.  The descriptor arrays are host arrays."""
        r0 = np.ascontiguousarray(seg_row0, dtype=np.int64)
        nw = np.ascontiguousarray(seg_nw, dtype=np.int32)
        if r0.shape != nw.shape or r0.ndim != 1:
            raise ValueError("seg_row0 and seg_nw must be 1-D arrays of the same length")
        self._chk(self._lib.ww_forward_segments_dev(self.ctx.handle, self._model, C.c_void_p(d_mel_ptr), int(mel_rows),
                                                    _lib.ptr(r0), _lib.ptr(nw), int(r0.size), int(hop), C.c_void_p(d_out_ptr)))


class StreamBank:
    """S device-resident streams advanced 20 ms per :meth:`step` (``ww_stream_*``)."""

    def __init__(self, engine: Engine, n_streams: int, fp: Optional[_lib.FrontendParams] = None,
                 full_recompute: bool = False, two_launch: bool = False, sync_wait: bool = False) -> None:
        """``full_recompute``: every streaming CRNN window recomputed from its mel rows (``WW_STREAM_FULL_RECOMPUTE``)
        instead of the incremental kernel.  ``two_launch``: the incremental CRNN's tick as a front-end kernel + a model kernel
        (``WW_STREAM_TWO_LAUNCH``; default: ONE launch per tick).  ``sync_wait``: wait for a tick with ``hipStreamSynchronize``
        instead of polling its posteriors in page-locked memory (``WW_STREAM_SYNC_WAIT``).  Same bits in every form."""
        self.engine = engine
        self.S = int(n_streams)
        self._lib = _lib.load()
        fp = fp or frontend_params()
        h = C.c_void_p()
        flags = ((_lib.STREAM_FULL_RECOMPUTE if full_recompute else 0) | (_lib.STREAM_TWO_LAUNCH if two_launch else 0)
                 | (_lib.STREAM_SYNC_WAIT if sync_wait else 0))
        _lib.raise_for(self._lib.ww_stream_create(engine.ctx.handle, engine.handle, self.S, C.byref(fp), flags, C.byref(h)),
                       engine.ctx.handle)
        self._h = h
        _lib.register("streams", self)
        self._post = np.zeros((self.S, 2), np.float32)
        self._n = np.zeros(self.S, np.int32)
        self._flags = np.zeros(self.S, np.uint8)
        # a tick is host-paced (spokestack/pipeline.py:25-28): the addresses of the bank's own arrays are taken once, not per call
        self._p_post, self._p_n, self._p_flags = (C.c_void_p(a.ctypes.data) for a in (self._post, self._n, self._flags))
        self._step = self._lib.ww_stream_step
        self._shape = (self.S, 320)
        self._keep = None

    def _frames_address(self, frames: np.ndarray) -> int:
        f = frames
        if not (type(f) is np.ndarray and f.dtype == np.int16 and f.flags.c_contiguous):
            f = self._keep = np.ascontiguousarray(frames, dtype=np.int16)  # (kept alive until the call has returned)
        if f.shape != self._shape:
            raise ValueError(f"frames must be [{self.S}, 320] int16")
        try:
            return C.addressof(C.c_char.from_buffer(f))  # (a third of the cost of f.ctypes.data_as)
        except (TypeError, ValueError):                   # a read-only array
            return f.ctypes.data

    def step_trigger(self, frames: np.ndarray, p_is_speech, p_is_active, threshold: float, p_state) -> None:
        """The wake-word stage of all streams as ONE library call (``ww_stream_step_trigger``): the tick, the trigger logic of
        ``WakewordTrigger.__call__`` over its posteriors and the reset of the streams whose VAD bit fell.  Every argument but
        ``frames`` is an address taken once with ``_lib.addr`` (``WakewordBank`` owns the arrays): ``p_state`` =
        (was_speech, posterior_max, post, n_post, fired_ids, n_fired, fall_ids, n_fall)."""
        rc = self._lib.ww_stream_step_trigger(self._h, self._frames_address(frames), p_is_speech, p_is_active, threshold, *p_state)
        if rc:
            _lib.raise_for(rc, self.engine.ctx.handle)

    def step(self, frames: np.ndarray, is_speech: np.ndarray, is_active: Optional[np.ndarray] = None) -> Tuple[np.ndarray, np.ndarray]:
        pf = self._frames_address(frames)
        sp = is_speech
        if not (type(sp) is np.ndarray and sp.dtype == np.uint8 and sp.ndim == 1):
            sp = np.ascontiguousarray(is_speech, dtype=np.uint8).ravel()
        if sp.size != self.S:
            raise ValueError("is_speech must have one entry per stream")
        np.bitwise_and(sp, 1, out=self._flags)
        if is_active is not None:
            self._flags |= (np.ascontiguousarray(is_active, dtype=np.uint8).ravel() & 1) << 1
        rc = self._step(self._h, pf, self._p_flags, self._p_post, self._p_n)
        if rc:
            _lib.raise_for(rc, self.engine.ctx.handle)
        return self._post.copy(), self._n.copy()  # the caller owns what it gets (like TFLiteModel's get_tensor copies)

    TIMELINE_PHASES = ("plan", "frames_in", "launch_1", "launch_2", "wait", "copy_out")

    def timeline(self, reset: bool = False) -> dict:
        """Mean host-side microseconds per phase of ``ww_stream_step`` since the last reset (``ww_stream_timeline``)."""
        ns = (C.c_double * len(self.TIMELINE_PHASES))()
        ticks = C.c_int64(0)
        _lib.raise_for(self._lib.ww_stream_timeline(self._h, ns, C.byref(ticks), int(reset)), self.engine.ctx.handle)
        out = {k: ns[i] * 1e-3 for i, k in enumerate(self.TIMELINE_PHASES)}
        out["ticks"] = int(ticks.value)
        return out

    def reset(self, ids: Optional[Sequence[int]] = None) -> None:
        if ids is None:
            _lib.raise_for(self._lib.ww_stream_reset(self._h, None, 0), self.engine.ctx.handle)
        else:
            a = np.ascontiguousarray(ids, dtype=np.int32)
            _lib.raise_for(self._lib.ww_stream_reset(self._h, _lib.ptr(a), a.size), self.engine.ctx.handle)

    def close(self) -> None:
        if self._h and not _lib.is_shutdown():
            self._lib.ww_stream_destroy(self._h)
        self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass
