"""Keyword recogniser stage with the surface of the reference's ``KeywordRecognizer``
(``spokestack/asr/keyword/tflite.py:15-195``).

Same front end as the wake-word plugin (normalise / 32767, clip, pre-emphasis with carry - default 0.97 here -,
512 / hop framing, Hann . rFFT . |.| in float64, ``filter.tflite``), which runs in the HIP front-end kernel; the frames
are only analysed while ``context.is_active`` although the sample ring always advances (``tflite.py:123-129``), the
encoder is autoregressive (``encode_model(frame_window, state) -> (encoded, state)``, ``:153-158``) and the detector runs
once, on the falling edge of ``is_active``, over the window of encoded frames (``:102-106,165-184``): arg-max class,
``recognize`` / ``timeout`` event, reset.

The reference ships NO keyword models (its ``model_dir`` is supplied by the user), so the encode / detect graphs cannot
be read into kernels here: they are plug-ins with ``TFLiteModel``'s call protocol and ``input_details`` /
``output_details`` (``CallableModel`` wraps a Python function; ``TFLiteModel`` is tried when none is given and raises
for graphs other than the wake-word architectures).  ``filter_model_dir`` names a model directory whose
``filter.tflite`` is the (identical) mel graph.
"""
from __future__ import annotations

import os
from typing import Callable, List, Optional, Sequence

import numpy as np

from .context import SpeechContext
from .engine import frontend_params
from .models import engine_for


class CallableModel:
    """A Python callable behind ``TFLiteModel``'s protocol: ``model(*arrays) -> list of arrays`` plus the
    ``input_details`` / ``output_details`` shape records the reference's constructors size their buffers from."""

    def __init__(self, fn: Callable[..., Sequence[np.ndarray]], input_shapes: Sequence[Sequence[int]],
                 output_shapes: Sequence[Sequence[int]]) -> None:
        self._fn = fn
        self.input_details = [{"shape": np.array(s, np.int32), "index": i} for i, s in enumerate(input_shapes)]
        self.output_details = [{"shape": np.array(s, np.int32), "index": i} for i, s in enumerate(output_shapes)]

    def __call__(self, *args: np.ndarray) -> List[np.ndarray]:
        return [np.asarray(o) for o in self._fn(*args)]


class KeywordRecognizer:
    def __init__(self, classes: List[str], pre_emphasis: float = 0.97, sample_rate: int = 16000, fft_window_type: str = "hann",
                 fft_hop_length: int = 10, model_dir: str = "", posterior_threshold: float = 0.5,
                 encode_model=None, detect_model=None, filter_model_dir: Optional[str] = None, device: int = 0,
                 **kwargs) -> None:
        self.classes = classes
        self.pre_emphasis = pre_emphasis
        self.hop_length = int(fft_hop_length * sample_rate / 1000)
        if fft_window_type != "hann":
            raise ValueError("Invalid fft_window_type")
        if encode_model is None or detect_model is None:
            from .models import TFLiteModel  # raises ValueError for graphs the kernels do not implement
            encode_model = encode_model or TFLiteModel(model_path=os.path.join(model_dir, "encode.tflite"))
            detect_model = detect_model or TFLiteModel(model_path=os.path.join(model_dir, "detect.tflite"))
        self.encode_model, self.detect_model = encode_model, detect_model
        if len(classes) != self.detect_model.output_details[0]["shape"][-1]:
            raise ValueError("Invalid number of classes")
        self._engine = engine_for(filter_model_dir or model_dir, device)   # the mel graph + the front-end kernel
        self._window_size = (self._engine.n_bins - 1) * 2
        self.mel_length = int(self.encode_model.input_details[0]["shape"][1])
        self.mel_width = int(self.encode_model.input_details[0]["shape"][-1])
        if self.mel_width != self._engine.n_mel:
            raise ValueError(f"encoder expects {self.mel_width} mel bands, the filter graph yields {self._engine.n_mel}")
        self.state = np.zeros(self.encode_model.input_details[1]["shape"], np.float32)
        self.encode_length = int(self.detect_model.input_details[0]["shape"][1])
        self.encode_width = int(self.detect_model.input_details[0]["shape"][-1])
        self._fp = frontend_params(1.0, False, 0.0, self.hop_length, True)
        self._pending = np.zeros(0, np.float32)          # the sample ring, as a flat tail
        self._frame_window = np.zeros((self.mel_length, self.mel_width), np.float32)           # filled with 0.0
        self._encode_window = np.full((self.encode_length, self.encode_width), -1.0, np.float32)  # filled with -1.0
        self._posterior_threshold = posterior_threshold
        self._prev_sample = 0.0
        self._is_active = False

    def __call__(self, context: SpeechContext, frame) -> None:
        self._sample(context, frame)
        if not context.is_active and self._is_active:
            self._detect(context)
        self._is_active = context.is_active

    def _sample(self, context: SpeechContext, frame) -> None:
        x = np.asarray(frame).astype(np.float32) / (2 ** 15 - 1)
        x = np.clip(x, -1.0, 1.0)
        prev = x[-1]
        x -= self.pre_emphasis * np.append(self._prev_sample, x[:-1])
        self._prev_sample = prev
        buf = np.concatenate((self._pending, x.astype(np.float32)))
        n = (len(buf) - self._window_size) // self.hop_length + 1 if len(buf) >= self._window_size else 0
        if n > 0:
            if context.is_active:  # constant during this call: only other stages change it
                used = self._window_size + (n - 1) * self.hop_length
                for mel in self._engine.logmel([buf[:used]], self._fp)[0]:
                    self._encode(mel)
            buf = buf[n * self.hop_length:]  # the ring advances whether or not the frames were analysed
        self._pending = buf.copy()

    def _encode(self, mel: np.ndarray) -> None:
        self._frame_window = np.roll(self._frame_window, -1, axis=0)
        self._frame_window[-1] = mel
        encoded, self.state = self.encode_model(self._frame_window[None], self.state)
        self._encode_window = np.roll(self._encode_window, -1, axis=0)
        self._encode_window[-1] = np.asarray(encoded, np.float32).reshape(-1)[: self.encode_width]

    def _detect(self, context: SpeechContext) -> None:
        posterior = np.asarray(self.detect_model(self._encode_window[None])[0][0])
        class_index = int(np.argmax(posterior))
        confidence = posterior[class_index]
        if confidence >= self._posterior_threshold:
            context.transcript = self.classes[class_index]
            context.confidence = confidence
            context.event("recognize")
        else:
            context.event("timeout")
        self.reset()

    def reset(self) -> None:
        self._pending = np.zeros(0, np.float32)
        self._frame_window[:] = 0.0
        self._encode_window[:] = -1.0
        self.state[:] = 0.0

    def close(self) -> None:
        self.reset()
