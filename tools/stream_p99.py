#!/usr/bin/env python3
"""Development: where do the slow ticks of the 128-stream bank come from?  p50/p99/max and the indices of the slowest ticks."""
import os, sys, time, gc
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np
from wwhip.engine import Engine, StreamBank
S, ticks = 128, int(sys.argv[1]) if len(sys.argv) > 1 else 5000
if "nogc" in sys.argv: gc.disable()
eng = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/CRNN"))
bank = StreamBank(eng, S)
rng = np.random.default_rng(0)
frames = np.clip(rng.normal(0, 2500, (64, S, 320)), -32768, 32767).astype(np.int16)
speech = np.ones(S, np.uint8)
for t in range(200): bank.step(frames[t % 64], speech)
lat = np.empty(ticks)
for t in range(ticks):
    t0 = time.perf_counter(); bank.step(frames[t % 64], speech); lat[t] = time.perf_counter() - t0
q = np.percentile(lat, [50, 90, 99, 99.9]) * 1e3
slow = np.argsort(lat)[-12:]
print("p50 %.4f p90 %.4f p99 %.4f p99.9 %.4f max %.4f ms" % (*q, lat.max() * 1e3), "slow ticks", sorted(slow.tolist()), "n>0.15ms", int((lat > 0.15e-3).sum()))
