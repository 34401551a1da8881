"""Deactivates the pipeline after a minimum/maximum activation length, like the reference's
``ActivationTimeout`` (``spokestack/activation_timeout.py:16-51``)."""
from __future__ import annotations

from .context import SpeechContext


class ActivationTimeout:
    def __init__(self, frame_width: int = 20, min_active: int = 500, max_active: int = 5000, **kwargs) -> None:
        self._min_frames = min_active / frame_width
        self._max_frames = max_active / frame_width
        self._was_speech = False
        self._active_frames = 0

    def __call__(self, context: SpeechContext, frame=None) -> None:
        fell = self._was_speech and not context.is_speech
        self._was_speech = context.is_speech
        if not context.is_active:
            return
        self._active_frames += 1
        if self._active_frames > self._min_frames and (fell or self._active_frames > self._max_frames):
            self.deactivate(context)

    def deactivate(self, context: SpeechContext) -> None:
        self.reset()
        context.is_active = False

    def reset(self) -> None:
        self._active_frames = 0

    def close(self) -> None:
        self._active_frames = 0


class ActivationTimeoutBank:
    """``ActivationTimeout`` for the S streams of a :class:`~wwhip.context.ContextBank` in lock step: one pass of the library's
    ``ww_timeout_bank_step`` per tick (spokestack/activation_timeout.py:25-38 per stream) over the bank's flag arrays;
    ``deactivate`` events are raised for the streams that timed out, and only for those."""

    def __init__(self, n_streams: int, frame_width: int = 20, min_active: int = 500, max_active: int = 5000, **kwargs) -> None:
        import numpy as np
        from . import _lib
        self.S = int(n_streams)
        self._min_frames = min_active / frame_width
        self._max_frames = max_active / frame_width
        self.was_speech = np.zeros(self.S, np.uint8)
        self.active_frames = np.zeros(self.S, np.int32)
        self._ids = np.zeros(self.S, np.int32)
        self._n = np.zeros(1, np.int32)
        self._fn = _lib.load().ww_timeout_bank_step
        self._p = tuple(_lib.addr(a) for a in (self.was_speech, self.active_frames, self._ids, self._n))
        self._bound = None  # (the ContextBank, the addresses of its two flag arrays)

    def __call__(self, contexts, frames=None) -> None:
        b = self._bound
        if b is None or b[0] is not contexts:
            from . import _lib
            if len(contexts) != self.S:
                raise ValueError("one context per stream")
            b = self._bound = (contexts, _lib.addr(contexts.is_speech), _lib.addr(contexts.is_active))
        p = self._p
        if self._fn(self.S, b[1], b[2], p[0], p[1], self._min_frames, self._max_frames, p[2], p[3]):
            raise ValueError("ww_timeout_bank_step refused its arguments")
        n = self._n[0]
        if n:
            contexts.emit("deactivate", self._ids[:n].copy())

    def reset(self) -> None:
        self.active_frames[:] = 0

    def close(self) -> None:
        self.active_frames[:] = 0
