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

    def __init__(self, input_source, stages: List, n_streams: int, fused: bool = True) -> None:
        """``fused``: when the stage list is exactly ``[VadBank, WakewordBank, ActivationTimeoutBank]`` (the stage list of the
        reference's ``demo.py``), a tick is ONE library call (``ww_pipeline_bank_step``: the three stages' passes in stage order on
        the stages' own arrays) instead of three - the same flags, events and posteriors; ``False`` keeps the stage-by-stage loop."""
        from .context import ContextBank
        super().__init__(input_source, stages)
        self._context = ContextBank(n_streams)
        self._fused = self._bind_fused() if fused else None

    def _bind_fused(self):
        """The state block of ``ww_pipeline_bank_step`` over the three stages' arrays, or ``None`` when the stages are not that trio."""
        from . import _lib
        from .activation_timeout import ActivationTimeoutBank
        from .engine import StreamBank
        from .vad import VadBank
        from .wakeword import WakewordBank
        st = self._stages
        if not (len(st) == 3 and type(st[0]) is VadBank and type(st[1]) is WakewordBank and type(st[2]) is ActivationTimeoutBank
                and type(st[1]._bank) is StreamBank and st[0]._classify is not None and st[0].S == st[1].S == st[2].S == len(self._context)):
            return None
        vad, wake, to, ctx = st[0], st[1], st[2], self._context
        a = _lib.addr
        ps = _lib.PipelineState()
        ps.is_speech, ps.is_active = a(ctx.is_speech), a(ctx.is_active)
        ps.raw, ps.run_value, ps.run_length = a(vad._raw), a(vad.run_value), a(vad.run_length)
        ps.wake_was_speech, ps.posterior_max, ps.post, ps.n_post = a(wake._was_speech), a(wake.posterior_max), a(wake._post), a(wake._n)
        ps.timeout_was_speech, ps.active_frames = a(to.was_speech), a(to.active_frames)
        ps.fired_ids, ps.fall_ids, ps.deact_ids = a(wake._fired), a(wake._fall), a(to._ids)
        ps.threshold, ps.min_frames, ps.max_frames = wake.threshold, to._min_frames, to._max_frames
        ps.rise_frames, ps.fall_frames = vad._rise, vad._fall
        import ctypes as C
        return (ps, C.byref(ps), _lib.load().ww_pipeline_bank_step, wake._bank, vad, wake)

    def cleanup(self) -> None:
        self._fused = None  # (the state block points into the stages' arrays: it goes before they do)
        super().cleanup()

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
        f = self._fused
        if f is None:
            for stage in self._stages:
                stage(self._context, frames)
            return
        ps, ref, fn, bank, vad, wake = f
        import numpy as np
        raw = vad._classify(frames)
        if np.shape(raw) != (vad.S,):
            raise ValueError(f"one raw decision per stream is needed ({vad.S}), got shape {np.shape(raw)}")
        np.copyto(vad._raw, raw, casting="unsafe")
        ps.threshold = wake.threshold  # (the one parameter a caller may change between ticks)
        rc = fn(bank._h, bank._frames_address(frames), ref)
        if rc:
            from . import _lib
            _lib.raise_for(rc, bank.engine.ctx.handle)
        if ps.n_fired:
            ids = wake._fired[:ps.n_fired].copy()
            if wake._on_wake is not None:
                wake._on_wake(ids)
            self._context.emit("activate", ids)
        if ps.n_deact:
            self._context.emit("deactivate", self._stages[2]._ids[:ps.n_deact].copy())
