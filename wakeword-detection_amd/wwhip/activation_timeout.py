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
