#!/usr/bin/env python3
"""Development: host timeline of the at-scale evaluation's staging pipeline (wwhip.evaluate._run_jobs) - when each chunk was
planned, submitted to the uploader and launched, on one clock.  usage: eval_trace.py [n_wake]"""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np
import torch
from wwhip import evaluate as E
from wwhip.models import engine_for
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2529
clips, labels = E.synth_testset_scaled(n, n)
eng = engine_for(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/CRNN_softmax"), 0)
log, T0 = [], [0.0]
ids = {}


def wrap(name, fn):
    def inner(*a, **k):
        ch = a[0] if name == "prep" else a[1]
        i = ids.setdefault(id(ch), len(ids))
        t0 = time.perf_counter()
        r = fn(*a, **k)
        log.append((name, i, t0 - T0[0], time.perf_counter() - T0[0], int(sum(ch.n_win))))
        return r
    return inner


E._prep_chunk = wrap("prep", E._prep_chunk)
E._submit_chunk = wrap("submit", E._submit_chunk)
E._chunk_forward = wrap("launch", E._chunk_forward)
sys.setswitchinterval(float(os.environ.get("WW_SWITCH", "0.005")))
for p in range(4):
    log.clear(); ids.clear()
    tm = {}
    torch.cuda.synchronize()
    T0[0] = time.perf_counter()
    r = E.evaluate_reference_flow_sharded(eng, clips, labels, timing=tm)
    el = time.perf_counter() - T0[0]
print(f"total {el * 1e3:.2f} ms  device {tm['device_ms']:.2f} ms  phases", {k: round(v * 1e3, 2) for k, v in tm.items() if isinstance(v, float) and k != "device_ms"})
for name, i, a, b, w in sorted(log, key=lambda x: x[2]):
    print(f"{name:7s} chunk {i:2d} ({w:6d} windows)  {a * 1e3:7.2f} -> {b * 1e3:7.2f} ms")
