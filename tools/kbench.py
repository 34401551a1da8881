#!/usr/bin/env python3
"""Per-kernel timing helper (development tool): runs the clips path under ww_profile and
prints average microseconds per kernel.  Usage: python tools/kbench.py [crnn|wavenet] [clips] [steps] [fast] [bf16x3]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np, torch
from wwhip.engine import Engine, frontend_params
from wwhip import _lib

model = sys.argv[1] if len(sys.argv) > 1 else "crnn"
clips = int(sys.argv[2]) if len(sys.argv) > 2 else 256
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 50
fast = "fast" in sys.argv[4:]
precision = "bf16x3" if "bf16x3" in sys.argv[4:] else "fp32"
ctx = _lib.Context(0)
eng = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models", "CRNN" if model == "crnn" else "Wavenet"), ctx=ctx, precision=precision)
rng = np.random.default_rng(0)
pcm = np.clip(rng.normal(0, 2000, (clips, 24000)), -32768, 32767).astype(np.int16)
d = torch.from_numpy(pcm).cuda(); out = torch.zeros((clips, eng.n_out), device="cuda")
fp = frontend_params(32767.0, True, 0.0, 160, not fast)
for _ in range(5): eng.clips_forward_dev(d.data_ptr(), clips, 24000, out.data_ptr(), fp)
ctx.synchronize()
ctx.profile(True)
for _ in range(steps): eng.clips_forward_dev(d.data_ptr(), clips, 24000, out.data_ptr(), fp)
p = ctx.profile_read(); ctx.profile(False)
print({k: round(v["total_ms"] / v["calls"] * 1e3, 2) for k, v in p.items()})
ctx.timer_start()
for _ in range(steps): eng.clips_forward_dev(d.data_ptr(), clips, 24000, out.data_ptr(), fp)
print("graph replay us/step", round(ctx.timer_stop() / steps * 1e3, 2))
