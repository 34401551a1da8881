#!/usr/bin/env python3
"""Development tool: host-to-host latency of the small per-frame calls the reference's plugin makes (one frame
through stft / filter, one window through encode+detect, detect alone, one clip through the front end)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np
from wwhip.engine import Engine
eng = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/CRNN"))
rng = np.random.default_rng(0)
frame = rng.normal(0, 0.1, (1, 512)).astype(np.float32)
mag = np.abs(rng.normal(0, 1, (1, 257))).astype(np.float32)
win = rng.normal(0, 1, (1, eng.window, 40)).astype(np.float32)
enc = rng.normal(0, 1, (1,) + tuple(eng.enc_shape)).astype(np.float32)
clip = [np.clip(rng.normal(0, 2000, 24000), -32768, 32767).astype(np.int16)]
cases = {"stft_mag(1 frame)": lambda: eng.stft_mag(frame), "filter_apply(1 frame)": lambda: eng.filter_apply(mag),
         "forward(1 window)": lambda: eng.forward(win), "detect(1)": lambda: eng.detect(enc), "logmel(1.5 s clip)": lambda: eng.logmel(clip)}
for name, fn in cases.items():
    for _ in range(20): fn()
    t0 = time.perf_counter()
    for _ in range(500): fn()
    print(f"{name:24s} {(time.perf_counter() - t0) / 500 * 1e6:7.1f} us")
