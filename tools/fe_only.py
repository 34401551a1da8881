#!/usr/bin/env python3
"""Front end alone (development tool): N launches of the batched log-mel kernel on resident PCM.
Usage: python tools/fe_only.py [clips] [launches] [fast]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np
from wwhip.engine import Engine, frontend_params
from wwhip import _lib
clips = int(sys.argv[1]) if len(sys.argv) > 1 else 256
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
fast = "fast" in sys.argv[3:]
ctx = _lib.Context(0)
eng = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/CRNN"), ctx=ctx)
rng = np.random.default_rng(0)
pcm = [np.clip(rng.normal(0, 2000, 24000), -32768, 32767).astype(np.int16) for _ in range(clips)]
fp = frontend_params(precise=not fast)
eng.logmel(pcm, fp)
ctx.profile(True)
for _ in range(n):
    eng.logmel(pcm, fp)
p = ctx.profile_read(); ctx.profile(False)
print({k: round(v["total_ms"] / v["calls"] * 1e3, 2) for k, v in p.items()})
