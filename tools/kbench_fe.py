#!/usr/bin/env python3
"""Front-end kernel timing (development tool)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np
from wwhip.engine import Engine, frontend_params
from wwhip import _lib
ctx = _lib.Context(0)
eng = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/CRNN"), ctx=ctx)
rng = np.random.default_rng(0)
frames = rng.normal(0, 0.1, (37632, 512)).astype(np.float32)
pcm = [np.clip(rng.normal(0, 2000, 24000), -32768, 32767).astype(np.int16) for _ in range(256)]
for precise in (True, False):
    eng.stft_mag(frames, precise); eng.logmel(pcm, frontend_params(precise=precise))
    ctx.profile(True)
    for _ in range(5):
        eng.stft_mag(frames, precise)
        eng.logmel(pcm, frontend_params(precise=precise))
    p = ctx.profile_read(); ctx.profile(False)
    print("precise", precise, {k: round(v["total_ms"] / v["calls"] * 1e3, 2) for k, v in p.items()})
