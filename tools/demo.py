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

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]

import numpy as np  # noqa: E402

from spokestack.activation_timeout import ActivationTimeout  # noqa: E402
from spokestack.pipeline import SpeechPipeline  # noqa: E402
from spokestack.vad.webrtc import VoiceActivityDetector  # noqa: E402
from spokestack.wakeword.tflite import WakewordTrigger  # noqa: E402
from spokestack.io.wav import WavInput  # noqa: E402
from wwhip.vad import EnergyClassifier  # noqa: E402

logging.basicConfig(level=logging.INFO)


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
    wake = WakewordTrigger(model_dir=args.models_dir, model_type=args.model_type, on_wake=lambda: wakes.append(mic.position_s))
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
