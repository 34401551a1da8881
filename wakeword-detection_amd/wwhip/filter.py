"""Offline mel front end with the surface of the reference's ``Filter``
(``utils/tf_lite/filter.py:9-79``): ``Filter(pre_emphasis, sample_rate, fft_window_type,
fft_hop_length, model_dir)``, ``filter_frame(frame) -> list of [40] arrays`` (possibly
empty), ``num_outputs()``.  State (pending samples, pre-emphasis carry) persists across calls
and across files exactly like the reference's never-reset sample ring.

The framing is closed form instead of a per-sample loop: with ``p`` pending samples and ``n``
new ones, ``max(0, (p + n - 512) // hop + 1)`` frames start at offsets ``0, hop, ...`` of the
concatenation and ``p + n - hop * frames`` samples stay pending.  Hann, FFT, mel and log run
in the HIP front-end kernel.
"""
from __future__ import annotations

from typing import List, Sequence

import numpy as np

from .engine import frontend_params
from .models import engine_for


class Filter:
    def __init__(self, pre_emphasis: float = 0.0, sample_rate: int = 16000, fft_window_type: str = "hann",
                 fft_hop_length: int = 10, model_dir: str = "", device: int = 0, precise: bool = True) -> None:
        self.pre_emphasis = pre_emphasis
        self.hop_length = int(fft_hop_length * sample_rate / 1000)
        if fft_window_type != "hann":
            raise ValueError("Invalid fft_window_type")
        self._engine = engine_for(model_dir, device)
        self._window_size = (self._engine.n_bins - 1) * 2
        self._pending = np.zeros(0, np.float32)
        self._prev_sample = 0.0
        self._fp = frontend_params(1.0, False, 0.0, self.hop_length, precise)

    # ---- reference surface ------------------------------------------------------------------
    def num_outputs(self) -> int:
        return self._engine.n_mel

    def filter_frame(self, frame: np.ndarray) -> List[np.ndarray]:
        feats = self.filter_samples(frame)
        return [row for row in feats]

    # ---- batch form ---------------------------------------------------------------------------
    def filter_samples(self, samples: np.ndarray) -> np.ndarray:
        """Any number of float samples -> ``[frames, 40]`` (continues the running stream)."""
        if len(samples) == 0:
            return np.zeros((0, self._engine.n_mel), np.float32)
        prev = samples[-1]
        if self.pre_emphasis != 0.0:
            # reference expression, in place on the caller's array (filter.py:43)
            samples -= self.pre_emphasis * np.append(self._prev_sample, samples[:-1])
        self._prev_sample = prev
        buf = np.concatenate((self._pending, np.asarray(samples, dtype=np.float32)))
        n_frames = (len(buf) - self._window_size) // self.hop_length + 1 if len(buf) >= self._window_size else 0
        if n_frames <= 0:
            self._pending = buf
            return np.zeros((0, self._engine.n_mel), np.float32)
        used = self._window_size + (n_frames - 1) * self.hop_length
        mel = self._engine.logmel([buf[:used]], self._fp)[0]
        self._pending = buf[n_frames * self.hop_length:].copy()
        return mel

    def reset(self) -> None:
        """Not in the reference (its ring is never reset); lets callers start a clean stream."""
        self._pending = np.zeros(0, np.float32)
        self._prev_sample = 0.0


def filter_utterances(engine, utterances: Sequence[np.ndarray], **fp_kwargs) -> List[np.ndarray]:
    """Independent utterances (ring reset per utterance) in ONE launch: list of int16/float32
    arrays -> list of ``[frames, 40]``."""
    return engine.logmel(list(utterances), frontend_params(**fp_kwargs))
