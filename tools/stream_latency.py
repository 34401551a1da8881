#!/usr/bin/env python3
"""BASELINE config 5 on one GPU: S concurrent 16 kHz streams, 20 ms per tick, is_speech = 1
(worst case: 2 posteriors per stream and tick).  Reports p50/p99 of the per-tick latency
(tick submitted on the host -> posteriors visible on the host)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np
from wwhip.engine import Engine, StreamBank
S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 10000  # SURVEY 8(d) cfg 5: 10,000 ticks after 100 warm-up
rng = np.random.default_rng(0)
out = {}
for name, prec in (("CRNN", "fp32"), ("Wavenet", "fp32"), ("Wavenet", "bf16x3")):
    eng = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models", name), precision=prec)
    bank = StreamBank(eng, S)
    frames = np.clip(rng.normal(0, 2500, (64, S, 320)), -32768, 32767).astype(np.int16)
    speech = np.ones(S, np.uint8)
    for t in range(100):
        bank.step(frames[t % 64], speech)
    lat = np.empty(ticks)
    n_post = 0
    for t in range(ticks):
        t0 = time.perf_counter()
        p, n = bank.step(frames[t % 64], speech)
        lat[t] = time.perf_counter() - t0
        n_post += int(n.sum())
    out[name if prec == "fp32" else name + "/" + prec] = {"streams": S, "ticks": ticks, "p50_ms": float(np.percentile(lat, 50) * 1e3),
                 "p99_ms": float(np.percentile(lat, 99) * 1e3), "mean_ms": float(lat.mean() * 1e3),
                 "posteriors_per_tick": n_post / ticks,
                 "realtime_factor": 0.020 / float(lat.mean())}
    bank.close(); eng.close()
print(json.dumps(out))
