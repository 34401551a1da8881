#!/usr/bin/env python3
"""Development: split-bf16 Wavenet against the CPU oracle (37 windows) and against the fp32 kernel (700 windows, twice),
e.g. for WWHIP_WV_NW variants of the kernel.  Usage: [WWHIP_WV_NW=4] python tools/wv_check.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np
from wwhip.engine import Engine
from oracle.cpu import CpuOracle
e = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/Wavenet"), precision="bf16x3")
o = CpuOracle(e.blob)
rng = np.random.default_rng(29)
wins = rng.uniform(0, 6.5, (37, e.window, 40)).astype(np.float32)
wins[3] = 0; wins[4, 100:] = 0
got, enc = e.forward(wins, want_enc=True)
want, wenc = o.forward(wins, want_enc=True)
print("posterior err", float(np.abs(got - want).max()), "encoder err", float(np.abs(enc - wenc).max()))
big = rng.uniform(0, 6.5, (700, e.window, 40)).astype(np.float32)
big[::7, 140:] = 0
e.set_precision("fp32"); ref = e.forward(big); e.set_precision("bf16x3")
for _ in range(2):
    print("700 windows vs fp32 kernel:", float(np.abs(e.forward(big) - ref).max()))
