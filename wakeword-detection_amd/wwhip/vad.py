"""Voice-activity stages with the surface of ``spokestack/vad/webrtc.py``.

The frame classifier of the reference is webrtcvad (a third-party C library, not installed here and
out of scope); everything around it - the run-length hysteresis of ``VoiceActivityDetector``
(``webrtc.py:52-77``) and ``VoiceActivityTrigger`` (``:88-110``) - is host logic and is mirrored
here.  The classifier is a plug-in ``classifier(frame_bytes, sample_rate) -> bool`` (webrtcvad's
``Vad.is_speech`` signature); when webrtcvad is importable it is the default, as in the reference.
``VadBank`` is the same hysteresis for S streams at once (one NumPy pass per tick), for use next
to ``WakewordBank``.
"""
from __future__ import annotations

import logging
from typing import Callable, Optional, Sequence

import numpy as np

from .context import SpeechContext

QUALITY = 0
LOW_BITRATE = 1
AGGRESSIVE = 2
VERY_AGGRESSIVE = 3

_LOG = logging.getLogger(__name__)


class EnergyClassifier:
    """A deliberately simple stand-in classifier (mean |x| above a threshold) for tests and demos.
    It is NOT the reference's VAD; use webrtcvad where that matters."""

    def __init__(self, threshold: float = 500.0) -> None:
        self.threshold = threshold

    def __call__(self, frame_bytes: bytes, sample_rate: int) -> bool:
        x = np.frombuffer(frame_bytes, dtype=np.int16)
        return bool(len(x)) and float(np.abs(x.astype(np.int32)).mean()) > self.threshold


def _default_classifier(mode: int) -> Callable[[bytes, int], bool]:
    try:
        import webrtcvad  # type: ignore
    except ImportError as e:
        raise RuntimeError("webrtcvad is not installed: pass classifier=callable(frame_bytes, sample_rate) -> bool "
                           "(e.g. wwhip.vad.EnergyClassifier())") from e
    return webrtcvad.Vad(mode).is_speech


class VoiceActivityDetector:
    """``spokestack/vad/webrtc.py:22-86`` - sets ``context.is_speech`` with rise / fall delays - as the one-stream case of
    :class:`VadBank` (defined below): the stage classifies its frame, the bank runs the hysteresis, the stage writes the
    context when the bank's output moved.  (``tests/test_host_logic.py`` replays a recorded trace of the reference class
    through it; ``tests/test_gpu_dropin.py`` holds 128 of these against one 128-stream bank tick by tick.)"""

    def __init__(self, sample_rate: int = 16000, frame_width: int = 20, vad_rise_delay: int = 0, vad_fall_delay: int = 0,
                 mode: int = QUALITY, classifier: Optional[Callable[[bytes, int], bool]] = None, **kwargs) -> None:
        self._sample_rate = sample_rate
        self._classify = classifier if classifier is not None else _default_classifier(mode)
        self._bank = VadBank(1, frame_width, vad_rise_delay, vad_fall_delay)

    def __call__(self, context: SpeechContext, frame: np.ndarray) -> None:
        self._bank.is_speech[0] = bool(context.is_speech)  # the context is the state; other stages may have written it
        now = bool(self._bank.step([self._classify(np.asarray(frame).tobytes(), self._sample_rate) > 0])[0])
        if now != bool(context.is_speech):
            context.is_speech = now
            _LOG.info("vad: %s", "true" if now else "false")

    def reset(self) -> None:
        self._bank.reset()

    def close(self) -> None:
        self.reset()


class VoiceActivityTrigger:
    """``spokestack/vad/webrtc.py:88-116``: activates the context on a rising speech edge."""

    def __init__(self) -> None:
        self._seen = False  # context.is_speech as of the previous frame

    def __call__(self, context: SpeechContext, frame: np.ndarray = None) -> None:
        rising = bool(context.is_speech) and not self._seen
        self._seen = bool(context.is_speech)
        if rising:
            context.is_active = True

    def reset(self) -> None:
        self._seen = False

    def close(self) -> None:
        self.reset()


class VadBank:
    """The detector's hysteresis for S streams in lock step: ``step(raw[S]) -> is_speech[S]``."""

    def __init__(self, n_streams: int, frame_width: int = 20, vad_rise_delay: int = 0, vad_fall_delay: int = 0) -> None:
        self.S = int(n_streams)
        self._rise = vad_rise_delay // frame_width
        self._fall = vad_fall_delay // frame_width
        self.run_value = np.zeros(self.S, bool)
        self.run_length = np.zeros(self.S, np.int64)
        self.is_speech = np.zeros(self.S, bool)

    def step(self, raw: Sequence[bool]) -> np.ndarray:
        raw = np.asarray(raw, bool)
        same = raw == self.run_value
        self.run_length = np.where(same, self.run_length + 1, 1)
        self.run_value = raw
        differs = self.run_value != self.is_speech
        rise = differs & self.run_value & (self.run_length >= self._rise)
        fall = differs & ~self.run_value & (self.run_length >= self._fall)
        self.is_speech = np.where(rise, True, np.where(fall, False, self.is_speech))
        return self.is_speech.copy()

    def reset(self, ids: Optional[Sequence[int]] = None) -> None:
        sel = slice(None) if ids is None else np.asarray(ids, np.int64)
        self.run_value[sel] = False
        self.run_length[sel] = 0
