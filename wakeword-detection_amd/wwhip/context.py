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
