"""Synchronous frame-dispatch loop with the surface of the reference's ``SpeechPipeline``
(``spokestack/pipeline.py:18-111``): ``start/stop/pause/resume/run/step/cleanup/event``,
stages are callables ``stage(context, frame)`` with ``close()``; one thread, list order."""
from __future__ import annotations

from typing import Callable, List, Optional

from .context import SpeechContext


class SpeechPipeline:
    def __init__(self, input_source, stages: List) -> None:
        self._context = SpeechContext()
        self._input_source = input_source
        self._stages = stages
        self._is_running = False
        self._is_paused = False

    # ---- control --------------------------------------------------------------------------------
    def start(self) -> None:
        if self._is_running:
            return
        self._input_source.start()
        self._is_running = True

    def stop(self) -> None:
        self._is_running = False
        self._input_source.stop()

    def close(self) -> None:
        self.stop()

    def pause(self) -> None:
        self._is_paused = True
        self._input_source.stop()

    def resume(self) -> None:
        self._is_paused = False
        self._input_source.start()

    def activate(self) -> None:
        self._context.is_active = True

    def deactivate(self) -> None:
        self._context.is_active = False

    # ---- loop -----------------------------------------------------------------------------------
    def step(self) -> None:
        self._context.event("step")
        if self._is_paused:
            return
        frame = self._input_source.read()
        for stage in self._stages:
            stage(self._context, frame)

    def run(self) -> None:
        while self._is_running:
            self.step()
        self.cleanup()

    def cleanup(self) -> None:
        for stage in self._stages:
            stage.close()
        self._stages.clear()
        self._input_source.close()
        self._input_source = None
        self._context.reset()

    # ---- events ---------------------------------------------------------------------------------
    def event(self, function: Optional[Callable] = None, name: Optional[str] = None):
        if function is None:
            return lambda fn: self.event(fn, name)
        self._context.add_handler(name or function.__name__.replace("on_", ""), function)
        return None

    @property
    def is_running(self) -> bool:
        return self._is_running

    @property
    def context(self) -> SpeechContext:
        return self._context


class SpeechPipelineBank(SpeechPipeline):
    """The same loop for S streams in lock step (BASELINE config 5): ``input_source.read()`` returns one 20 ms frame per stream
    (``int16[S, 320]``), the context is a :class:`~wwhip.context.ContextBank`, the stages are the banked ones -
    ``VadBank``, ``WakewordBank``, ``ActivationTimeoutBank`` - called as ``stage(contexts, frames)`` in list order, one call per
    stage and tick whatever S is.  ``pipeline.context[s]`` is stream ``s``'s ``SpeechContext``; ``pipeline.event`` registers a
    handler for every stream (it receives the stream's context)."""

    def __init__(self, input_source, stages: List, n_streams: int) -> None:
        from .context import ContextBank
        super().__init__(input_source, stages)
        self._context = ContextBank(n_streams)

    def activate(self) -> None:
        for c in self._context:
            c.is_active = True

    def deactivate(self) -> None:
        for c in self._context:
            c.is_active = False

    def step(self) -> None:
        self._context.event("step")  # (spokestack/pipeline.py:26; once per tick, the handler receives the ContextBank)
        if self._is_paused:
            return
        frames = self._input_source.read()
        for stage in self._stages:
            stage(self._context, frames)
