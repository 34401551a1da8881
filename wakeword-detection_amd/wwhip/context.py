"""Shared pipeline state with the surface of the reference's ``SpeechContext``
(``spokestack/context.py:12-128``): ``is_speech``, ``is_active`` (edge-triggered
``activate`` / ``deactivate`` events), ``transcript``, ``confidence``, named handlers."""
from __future__ import annotations

import logging
from typing import Callable, Dict

_LOG = logging.getLogger(__name__)


class SpeechContext:
    def __init__(self) -> None:
        self._handlers: Dict[str, Callable] = {}
        self._state = {"is_speech": False, "is_active": False, "transcript": "", "confidence": 0.0}

    # ---- events ---------------------------------------------------------------------------------
    def add_handler(self, name: str, function: Callable) -> None:
        self._handlers[name] = function

    def event(self, name: str) -> None:
        fn = self._handlers.get(name)
        if fn is not None:
            fn(self)

    # ---- state ----------------------------------------------------------------------------------
    @property
    def is_speech(self) -> bool:
        return self._state["is_speech"]

    @is_speech.setter
    def is_speech(self, value: bool) -> None:
        self._state["is_speech"] = value

    @property
    def is_active(self) -> bool:
        return self._state["is_active"]

    @is_active.setter
    def is_active(self, value: bool) -> None:
        was = self._state["is_active"]
        self._state["is_active"] = value
        if bool(value) != bool(was):
            name = "activate" if value else "deactivate"
            self.event(name)
            _LOG.info("%s event", name)

    @property
    def transcript(self) -> str:
        return self._state["transcript"]

    @transcript.setter
    def transcript(self, value: str) -> None:
        self._state["transcript"] = value

    @property
    def confidence(self) -> float:
        return self._state["confidence"]

    @confidence.setter
    def confidence(self, value: float) -> None:
        self._state["confidence"] = value

    def reset(self) -> None:
        self.is_speech = False
        self.is_active = False
        self.transcript = ""
        self.confidence = 0.0


class _ContextView(SpeechContext):
    """Stream ``s`` of a :class:`ContextBank` with the whole ``SpeechContext`` surface: the two flags live in the bank's arrays
    (so a stage that works on all streams at once and a handler that holds this object see the same state), handlers, transcript and
    confidence are this stream's own."""

    def __init__(self, bank: "ContextBank", s: int) -> None:
        super().__init__()
        self._bank = bank
        self._s = s

    @property
    def is_speech(self) -> bool:
        return bool(self._bank.is_speech[self._s])

    @is_speech.setter
    def is_speech(self, value: bool) -> None:
        self._bank.is_speech[self._s] = bool(value)

    @property
    def is_active(self) -> bool:
        return bool(self._bank.is_active[self._s])

    @is_active.setter
    def is_active(self, value: bool) -> None:
        was = bool(self._bank.is_active[self._s])
        self._bank.is_active[self._s] = bool(value)
        if bool(value) != was:  # spokestack/context.py:71-85: the event fires on the edge
            self._bank.emit("activate" if value else "deactivate", (self._s,))


class ContextBank:
    """The ``SpeechContext`` of S streams in lock step (BASELINE config 5): ``is_speech[S]`` and ``is_active[S]`` as uint8 arrays
    that the banked stages (``VadBank``, ``WakewordBank``, ``ActivationTimeoutBank``) read and update in place in one pass per
    tick, and ``bank[s]`` as a ``SpeechContext`` *view* of stream ``s`` for handlers and single-stream stages.  A stage that
    changed flags calls :meth:`emit` with the ids that changed: activate / deactivate events reach exactly those streams'
    handlers (spokestack/context.py:71-85), a tick in which nothing changes costs no per-stream Python at all."""

    def __init__(self, n_streams: int) -> None:
        import numpy as np
        self.S = int(n_streams)
        self.is_speech = np.zeros(self.S, np.uint8)
        self.is_active = np.zeros(self.S, np.uint8)
        self._views: Dict[int, _ContextView] = {}
        self._handlers: Dict[str, Callable] = {}

    def __len__(self) -> int:
        return self.S

    def __getitem__(self, s: int) -> _ContextView:
        if not 0 <= s < self.S:
            raise IndexError(s)
        v = self._views.get(s)
        if v is None:
            v = self._views[s] = _ContextView(self, s)
        return v

    def __iter__(self):
        return (self[s] for s in range(self.S))

    def add_handler(self, name: str, function: Callable) -> None:
        """A bank-wide handler ``function(view)`` called for every stream the event reaches (after the stream's own)."""
        self._handlers[name] = function

    def event(self, name: str) -> None:
        """An event of the whole bank (the pipeline's per-tick ``"step"``): the bank-wide handler, called with the bank itself."""
        fn = self._handlers.get(name)
        if fn is not None:
            fn(self)

    def emit(self, name: str, ids) -> None:
        """Raise event ``name`` for the streams ``ids`` (whose flags the caller has already written)."""
        fn = self._handlers.get(name)
        info = _LOG.isEnabledFor(logging.INFO)
        for s in ids:
            s = int(s)
            v = self._views.get(s)
            if v is not None:
                v.event(name)
            if fn is not None:
                fn(self[s])
            if info:
                _LOG.info("%s event (stream %d)", name, s)

    def reset(self) -> None:
        active = self.is_active.nonzero()[0]
        self.is_speech[:] = 0
        self.is_active[:] = 0
        self.emit("deactivate", active)
        for v in self._views.values():
            v.transcript = ""
            v.confidence = 0.0
