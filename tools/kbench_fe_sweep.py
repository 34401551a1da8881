#!/usr/bin/env python3
"""Front-end kernel time against batch size (development tool): shows the block-round quantisation."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np
from wwhip.engine import Engine, frontend_params
from wwhip import _lib
ctx = _lib.Context(0)
eng = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/CRNN"), ctx=ctx)
rng = np.random.default_rng(0)
allpcm = [np.clip(rng.normal(0, 2000, 24000), -32768, 32767).astype(np.int16) for _ in range(512)]
for precise in (True, False):
    for n in [int(a) for a in sys.argv[1:]] or [64, 128, 192, 230, 256, 307, 384, 512]:
        pcm = allpcm[:n]
        eng.logmel(pcm, frontend_params(precise=precise))
        ctx.profile(True)
        for _ in range(8):
            eng.logmel(pcm, frontend_params(precise=precise))
        p = ctx.profile_read(); ctx.profile(False)
        k = "logmel_kernel<f64>" if precise else "logmel_kernel<f32>"
        us = p[k]["total_ms"] / p[k]["calls"] * 1e3
        print(f"precise={precise} clips={n} blocks={n*10} us={us:.2f} ns/clip={us*1e3/n:.1f}", flush=True)
