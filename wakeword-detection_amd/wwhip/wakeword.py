"""Streaming wake-word stage with the surface of the reference's ``WakewordTrigger``
(``spokestack/wakeword/tflite.py:20-250``).  Per 20 ms frame the reference normalises,
pre-emphasises, pushes 320 samples through a Python ring, and for every completed STFT frame
(while ``context.is_speech``) runs filter -> slide window -> encode -> detect.  Here all of
that state lives on the GPU in a one-stream :class:`StreamBank`; this class keeps only the
host logic: VAD-edge reset, activation, running maximum.

:class:`WakewordBank` is the many-stream form used by BASELINE config 5.
"""
from __future__ import annotations

import logging
from typing import Callable, Optional, Sequence

import numpy as np

from .context import SpeechContext
from .engine import StreamBank, frontend_params
from .models import engine_for

_LOG = logging.getLogger(__name__)


class WakewordTrigger:
    def __init__(self, pre_emphasis: float = 0.0, sample_rate: int = 16000, fft_window_type: str = "hann",
                 fft_hop_length: int = 10, model_dir: str = "", model_type: str = "",
                 posterior_threshold: float = 0.5, on_wake: Optional[Callable[[], None]] = None, device: int = 0,
                 superframe_len: int = 0, **kwargs) -> None:
        self.pre_emphasis = pre_emphasis
        self.hop_length = int(fft_hop_length * sample_rate / 1000)
        if fft_window_type != "hann":
            raise ValueError("Invalid fft_window_type")
        self.model_type = model_type.upper()
        if self.model_type not in ("CRNN", "WAVENET"):
            # the reference fails later with AttributeError; fail early instead
            raise ValueError(f"model_type must be 'CRNN' or 'Wavenet', got {model_type!r}")
        self._engine = engine_for(model_dir, device)
        if self._engine.is_crnn != (self.model_type == "CRNN"):
            raise ValueError(f"model_dir holds a {'CRNN' if self._engine.is_crnn else 'Wavenet'} model, "
                             f"model_type says {model_type}")
        self._window_size = (self._engine.n_bins - 1) * 2
        self.mel_length = self._engine.window
        self.mel_width = self._engine.n_mel
        self.encode_length, self.encode_width = self._engine.enc_shape
        self._bank = StreamBank(self._engine, 1, frontend_params(32767.0, True, pre_emphasis, self.hop_length, True))
        self._posterior_threshold = posterior_threshold
        self._posterior_max = 0.0
        self._is_speech = False
        self._on_wake = on_wake  # replaces the pydub audio reply (tflite.py:111-121,238)
        # superframe_len > 0 selects the experimental trigger of utils/CRNN_files/tflite.py:252-263
        # (shortest path over a superframe of posteriors instead of the 0.5 threshold)
        from .wfst import SuperframeDetector
        self._superframe = SuperframeDetector(superframe_len) if superframe_len > 0 else None

    def __call__(self, context: SpeechContext, frame) -> None:
        vad_fall = self._is_speech and not context.is_speech
        self._is_speech = context.is_speech
        if not context.is_active:
            self._sample(context, frame)
        if vad_fall:
            if not context.is_active:
                _LOG.info(f"wake: {self._posterior_max}")
            self.reset()

    def _sample(self, context: SpeechContext, frame) -> None:
        f = np.asarray(frame, dtype=np.int16).reshape(1, -1)
        if f.shape[1] != 320:
            raise ValueError("WakewordTrigger expects 20 ms frames of 320 int16 samples")
        post, n = self._bank.step(f, np.array([1 if context.is_speech else 0], np.uint8))
        for k in range(int(n[0])):
            posterior = float(post[0, k])
            if posterior > self._posterior_max:
                self._posterior_max = posterior
            if self._superframe is not None:
                fire = self._superframe.push(posterior)
            else:
                fire = posterior > self._posterior_threshold
            if fire and not context.is_active:
                _LOG.info(f"AWAKE!: {self._posterior_max}")
                if self._on_wake is not None:
                    self._on_wake()
                context.is_active = True

    def reset(self) -> None:
        self._bank.reset()
        self._posterior_max = 0.0
        if self._superframe is not None:
            self._superframe.reset()

    def close(self) -> None:
        self.reset()


class WakewordBank:
    """S streams in lock step: the batched form of :class:`WakewordTrigger` - one kernel launch per 20 ms tick for all streams,
    and the trigger's host logic (VAD-edge reset, running maximum, threshold, activation; tflite.py:134-146,232-239) as one pass
    over arrays inside the same library call (``ww_stream_step_trigger``).

    ``contexts`` is a :class:`~wwhip.context.ContextBank` - the stage then reads and writes the bank's flag arrays in place and
    raises ``activate`` events for the streams that fired, no per-stream Python otherwise - or, as before, a sequence of S
    ``SpeechContext`` objects (their flags are gathered and written back one by one: the slow form)."""

    def __init__(self, n_streams: int, model_dir: str = "", posterior_threshold: float = 0.5, pre_emphasis: float = 0.0,
                 device: int = 0, on_wake: Optional[Callable[[np.ndarray], None]] = None, bank=None) -> None:
        from . import _lib
        self.S = int(n_streams)
        if bank is None:
            if not model_dir:
                raise ValueError("WakewordBank needs model_dir (or bank=: a StreamBank built on an Engine of the caller's)")
            self._engine = engine_for(model_dir, device)
            bank = StreamBank(self._engine, self.S, frontend_params(32767.0, True, pre_emphasis, 160, True))
        self._bank = bank  # (anything with StreamBank's step_trigger / reset / close: the host-only tests pass a stub)
        self.threshold = float(posterior_threshold)
        self._on_wake = on_wake  # called with the ids of the streams that woke up this tick
        self.posterior_max = np.zeros(self.S, np.float32)
        self._was_speech = np.zeros(self.S, np.uint8)
        self._post = np.zeros((self.S, 2), np.float32)
        self._n = np.zeros(self.S, np.int32)
        self._fired = np.zeros(self.S, np.int32)
        self._fall = np.zeros(self.S, np.int32)
        self._counts = np.zeros(2, np.int32)
        self._p_state = (_lib.addr(self._was_speech), _lib.addr(self.posterior_max), _lib.addr(self._post), _lib.addr(self._n),
                         _lib.addr(self._fired), _lib.addr(self._counts[0:1]), _lib.addr(self._fall), _lib.addr(self._counts[1:2]))
        self._bound = None     # (the ContextBank, the addresses of its two flag arrays)
        self._scratch = None   # flag arrays for the sequence-of-contexts form

    @property
    def n_post(self) -> np.ndarray:
        """Posteriors per stream delivered by the last tick (0, 1 or 2)."""
        return self._n

    def step(self, contexts, frames: np.ndarray) -> np.ndarray:
        from . import _lib
        if hasattr(contexts, "emit"):  # a ContextBank: its arrays are the state
            b = self._bound
            if b is None or b[0] is not contexts:
                if len(contexts) != self.S:
                    raise ValueError("one context per stream")
                b = self._bound = (contexts, _lib.addr(contexts.is_speech), _lib.addr(contexts.is_active))
            self._bank.step_trigger(frames, b[1], b[2], self.threshold, self._p_state)
            nf = self._counts[0]
            if nf:
                ids = self._fired[:nf].copy()
                if self._on_wake is not None:
                    self._on_wake(ids)
                contexts.emit("activate", ids)
            return self._post.copy()
        # a sequence of SpeechContext objects
        if len(contexts) != self.S:
            raise ValueError("one context per stream")
        sc = self._scratch
        if sc is None:
            speech, active = np.zeros(self.S, np.uint8), np.zeros(self.S, np.uint8)
            sc = self._scratch = (speech, active, _lib.addr(speech), _lib.addr(active))
        sc[0][:] = [c.is_speech for c in contexts]
        sc[1][:] = [c.is_active for c in contexts]
        self._bank.step_trigger(frames, sc[2], sc[3], self.threshold, self._p_state)
        nf = self._counts[0]
        if nf:
            ids = self._fired[:nf].copy()
            if self._on_wake is not None:
                self._on_wake(ids)
            for s in ids:
                contexts[int(s)].is_active = True  # (the setter raises the activate event)
        return self._post.copy()

    __call__ = step  # the stage form: bank(contexts, frames)

    def reset(self) -> None:
        self._bank.reset()
        self.posterior_max[:] = 0.0

    def close(self) -> None:
        self._bank.close()
