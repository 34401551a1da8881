#!/usr/bin/env python3
"""demo.py of the reference (VAD -> wake word -> activation timeout in a SpeechPipeline) with a wav file in
place of the microphone: PyAudio and webrtcvad are not available here, so the input stage reads 20 ms frames from
``--wav`` and the VAD classifier is webrtcvad when installed, an energy threshold otherwise.

    python tools/demo.py --models_dir <dir> --model_type CRNN --wav some_16k_mono.wav"""
import argparse
import logging
import os
import sys
import time
import wave

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]

import numpy as np  # noqa: E402

from spokestack.activation_timeout import ActivationTimeout  # noqa: E402
from spokestack.pipeline import SpeechPipeline  # noqa: E402
from spokestack.vad.webrtc import VoiceActivityDetector  # noqa: E402
from spokestack.wakeword.tflite import WakewordTrigger  # noqa: E402
from wwhip.vad import EnergyClassifier  # noqa: E402

logging.basicConfig(level=logging.INFO)


class WavInput:
    """Input stage with PyAudioInput's surface (``read() -> int16 frame``, ``start/stop/close``); stops the
    pipeline at end of file."""

    def __init__(self, path: str, sample_rate: int = 16000, frame_width: int = 20) -> None:
        with wave.open(path, "rb") as w:
            if w.getframerate() != sample_rate or w.getsampwidth() != 2 or w.getnchannels() != 1:
                raise ValueError(f"{path}: expected mono PCM16 at {sample_rate} Hz")
            self._pcm = np.frombuffer(w.readframes(w.getnframes()), np.int16)
        self._n = sample_rate // 1000 * frame_width
        self._pos = 0
        self.pipeline = None

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


def parse_args():
    p = argparse.ArgumentParser(description="Spokestack demo script for VAD and Wake Word detection")
    p.add_argument("--models_dir", type=str, default=os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/CRNN"),
                   help="directory with TF-Lite models filter, decode, detect")
    p.add_argument("--model_type", type=str, default="CRNN", choices=["CRNN", "Wavenet"], help="Model architecture, Conv RNN, or Wavenet")
    p.add_argument("--sr", type=int, default=16000, help="Sample rate for audio (Hz)")
    p.add_argument("--fw", type=int, default=20, help="Frame width for audio in (ms)")
    p.add_argument("--wav", type=str, required=True, help="16 kHz mono PCM16 wav read instead of the microphone")
    return p.parse_args()


def main(args):
    start = time.time()
    mic = WavInput(args.wav, sample_rate=args.sr, frame_width=args.fw)
    try:
        vad = VoiceActivityDetector()
    except RuntimeError:
        vad = VoiceActivityDetector(classifier=EnergyClassifier(500.0))
    wakes = []
    wake = WakewordTrigger(model_dir=args.models_dir, model_type=args.model_type, on_wake=lambda: wakes.append(mic._pos / args.sr))
    timeout = ActivationTimeout(frame_width=args.fw)
    pipeline = SpeechPipeline(mic, [vad, wake, timeout])
    mic.pipeline = pipeline
    pipeline.start()
    pipeline.run()
    print(f"wake events at (s): {[round(t, 2) for t in wakes]}")
    print(f"Script completed in {time.time() - start:.2f} secs")
    return 0


if __name__ == "__main__":
    sys.exit(main(parse_args()))
