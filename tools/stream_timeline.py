#!/usr/bin/env python3
"""BASELINE config 5, one tick taken apart: where the per-tick latency of `StreamBank.step` goes OUTSIDE the kernels.

Per model: p50 / p99 of the whole `bank.step` call (what bench.py's streaming leg reports), the mean of the C entry
point's own phases (`ww_stream_timeline`: plan, frames in, launch 1, launch 2, wait, copy-out), and what is left for the
Python wrapper (argument marshalling, the ctypes call, the two result copies) = step mean - sum of the C phases.
Usage: stream_timeline.py [streams=128] [ticks=10000] [out.json]"""
import json
import os
import sys
import time

ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np  # noqa: E402
from wwhip.engine import Engine, StreamBank  # noqa: E402

S = int(sys.argv[1]) if len(sys.argv) > 1 else 128
ticks = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
rng = np.random.default_rng(5)
frames = np.clip(rng.normal(0, 2500, (64, S, 320)), -32768, 32767).astype(np.int16)
speech = np.ones(S, np.uint8)
out = {"streams": S, "ticks": ticks, "unit": "us"}
for name, prec, kw in (("CRNN", "fp32", {}), ("CRNN", "fp32", {"full_recompute": True}), ("Wavenet", "bf16x3", {}), ("Wavenet", "fp32", {})):
    eng = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models", name), precision=prec)
    bank = StreamBank(eng, S, **kw)
    for t in range(200):
        bank.step(frames[t % 64], speech)
    bank.timeline(reset=True)
    lat = np.empty(ticks)
    for t in range(ticks):
        t0 = time.perf_counter()
        bank.step(frames[t % 64], speech)
        lat[t] = time.perf_counter() - t0
    tl = bank.timeline()
    n = tl.pop("ticks")
    c_sum = sum(tl.values())
    key = name.lower() + ("" if prec == "fp32" else "_" + prec) + ("_full_recompute" if kw else "")
    out[key] = {"step_p50": float(np.percentile(lat, 50) * 1e6), "step_p99": float(np.percentile(lat, 99) * 1e6),
                "step_mean": float(lat.mean() * 1e6), "c_phases_mean": {k: round(v, 3) for k, v in tl.items()},
                "c_total_mean": c_sum, "python_wrapper_mean": float(lat.mean() * 1e6) - c_sum, "c_ticks": n}
    bank.close()
    eng.close()
text = json.dumps(out, indent=1)
print(text)
if len(sys.argv) > 3:
    with open(sys.argv[3], "w") as f:
        f.write(text + "\n")
