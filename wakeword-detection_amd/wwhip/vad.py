"""Voice-activity stages with the surface of ``spokestack/vad/webrtc.py``.

The frame classifier of the reference is webrtcvad (a third-party C library, not installed here and
out of scope); everything around it - the run-length hysteresis of ``VoiceActivityDetector``
(``webrtc.py:52-77``) and ``VoiceActivityTrigger`` (``:88-110``) - is host logic and is mirrored
here.  The classifier is a plug-in ``classifier(frame_bytes, sample_rate) -> bool`` (webrtcvad's
``Vad.is_speech`` signature); when webrtcvad is importable it is the default, as in the reference.
``VadBank`` is the same hysteresis for S streams at once (one pass of the library's ``ww_vad_bank_step``
per tick), for use next to ``WakewordBank`` and ``ActivationTimeoutBank`` on a ``ContextBank``.
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
    """The detector's hysteresis for S streams in lock step, one pass of the library's ``ww_vad_bank_step`` per tick
    (spokestack/vad/webrtc.py:59-77 per stream).  Two forms: ``step(raw[S]) -> is_speech[S]`` on the bank's own ``is_speech``
    array, and the pipeline-stage form ``bank(contexts, frames)`` where ``contexts`` is a :class:`~wwhip.context.ContextBank`
    whose ``is_speech`` array IS the state (as the context is in the reference) and the raw decisions come from a batch
    classifier ``classifier(frames[S, 320]) -> bool[S]`` (or are passed as ``raw=``)."""

    def __init__(self, n_streams: int, frame_width: int = 20, vad_rise_delay: int = 0, vad_fall_delay: int = 0,
                 classifier: Optional[Callable[[np.ndarray], Sequence[bool]]] = None) -> None:
        from . import _lib
        self.S = int(n_streams)
        self._rise = vad_rise_delay // frame_width
        self._fall = vad_fall_delay // frame_width
        self._classify = classifier
        self.run_value = np.zeros(self.S, np.uint8)
        self.run_length = np.zeros(self.S, np.int64)
        self.is_speech = np.zeros(self.S, np.uint8)
        self._raw = np.zeros(self.S, np.uint8)
        self._n_changed = np.zeros(1, np.int32)
        self._fn = _lib.load().ww_vad_bank_step
        self._p = tuple(_lib.addr(a) for a in (self._raw, self.run_value, self.run_length, self.is_speech, self._n_changed))
        self._bound = None  # the ContextBank whose is_speech array the stage form works on, and that array's address

    def _run(self, raw, p_speech) -> int:
        if np.shape(raw) != (self.S,):
            raise ValueError(f"one raw decision per stream is needed ({self.S}), got shape {np.shape(raw)}")
        np.copyto(self._raw, raw, casting="unsafe")
        p = self._p
        if self._fn(self.S, p[0], self._rise, self._fall, p[1], p[2], p_speech, p[4]):
            raise ValueError("ww_vad_bank_step refused its arguments")
        return int(self._n_changed[0])

    def step(self, raw: Sequence[bool]) -> np.ndarray:
        self._run(raw, self._p[3])
        return self.is_speech.astype(bool)

    def __call__(self, contexts, frames: Optional[np.ndarray] = None, raw: Optional[Sequence[bool]] = None) -> None:
        if self._bound is None or self._bound[0] is not contexts:
            from . import _lib
            if len(contexts) != self.S:
                raise ValueError("one context per stream")
            self._bound = (contexts, _lib.addr(contexts.is_speech))
        if raw is None:
            if self._classify is None:
                raise ValueError("VadBank needs classifier=callable(frames[S, 320]) -> bool[S], or raw= per call")
            raw = self._classify(frames)
        if self._run(raw, self._bound[1]) and _LOG.isEnabledFor(logging.INFO):
            _LOG.info("vad: %d streams speaking", int(contexts.is_speech.sum()))

    def reset(self, ids: Optional[Sequence[int]] = None) -> None:
        sel = slice(None) if ids is None else np.asarray(ids, np.int64)
        self.run_value[sel] = 0
        self.run_length[sel] = 0

    def close(self) -> None:
        self.reset()
