#!/usr/bin/env python3
"""Development: tick latency of the one-launch and the two-launch form of a bank by stream count (which form should a bank of S
streams take?).  usage: stream_forms.py [ticks=2000] [S ...]"""
import os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np
from wwhip.engine import Engine, StreamBank
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
sizes = [int(x) for x in sys.argv[2:]] or [128, 192, 256, 384, 512, 1024]
rng = np.random.default_rng(5)
for name, prec in (("CRNN", "fp32"), ("Wavenet", "bf16x3"), ("Wavenet", "fp32")):
    eng = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models", name), precision=prec)
    for S in sizes:
        frames = np.clip(rng.normal(0, 2500, (16, S, 320)), -32768, 32767).astype(np.int16)
        speech = np.ones(S, np.uint8)
        row = []
        for kw in ({}, {"two_launch": True}):
            bank = StreamBank(eng, S, **kw)
            for t in range(100):
                bank.step(frames[t % 16], speech)
            lat = np.empty(ticks)
            for t in range(ticks):
                t0 = time.perf_counter()
                bank.step(frames[t % 16], speech)
                lat[t] = time.perf_counter() - t0
            row.append(float(np.percentile(lat, 50) * 1e6))
            bank.close()
        print(f"{name:8s} {prec:7s} S={S:5d}  one launch p50 {row[0]:7.1f} us   two launches p50 {row[1]:7.1f} us", flush=True)
    eng.close()
