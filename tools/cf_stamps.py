#!/usr/bin/env python3
"""Development: one launch of crnn_fused_kernel over N windows with WWHIP_CF_STAMPS=1 -> the s_memtime stamps of the
phase boundaries (stderr, printed by the library).  Usage: WWHIP_CF_STAMPS=1 python tools/cf_stamps.py [windows] [fp32|bf16x3]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np
from wwhip.engine import Engine
n = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eng = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/CRNN"), precision=sys.argv[2] if len(sys.argv) > 2 else "fp32")
wins = np.random.default_rng(0).uniform(0, 6, (n, 151, 40)).astype(np.float32)
for i in range(3):
    print("--- launch", i, file=sys.stderr)
    eng.forward(wins)
