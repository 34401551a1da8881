"""Audio input stages with the surface of ``spokestack/io/pyaudio.py`` (``read() -> int16 frame``, ``start``,
``stop``, ``close``).  PyAudio and a microphone are not part of the hot path; :class:`WavInput` feeds a pipeline
from a 16 kHz mono PCM16 wav file (or an array) and stops it at the end of the data."""
from __future__ import annotations

import wave
from typing import Union

import numpy as np


class WavInput:
    def __init__(self, source: Union[str, np.ndarray], sample_rate: int = 16000, frame_width: int = 20) -> None:
        if isinstance(source, str):
            with wave.open(source, "rb") as w:
                if w.getframerate() != sample_rate or w.getsampwidth() != 2 or w.getnchannels() != 1:
                    raise ValueError(f"{source}: expected mono PCM16 at {sample_rate} Hz")
                self._pcm = np.frombuffer(w.readframes(w.getnframes()), np.int16)
        else:
            self._pcm = np.ascontiguousarray(source, dtype=np.int16)
        self._n = sample_rate // 1000 * frame_width
        self._pos = 0
        self.sample_rate = sample_rate
        self.pipeline = None  # set to the SpeechPipeline to have it stopped at end of data

    @property
    def position_s(self) -> float:
        return self._pos / self.sample_rate

    def read(self) -> np.ndarray:
        fr = self._pcm[self._pos:self._pos + self._n]
        self._pos += self._n
        if len(fr) < self._n:
            fr = np.pad(fr, (0, self._n - len(fr)))
            if self.pipeline is not None:
                self.pipeline.stop()
        return fr

    def start(self) -> None:
        pass

    def stop(self) -> None:
        pass

    def close(self) -> None:
        pass
