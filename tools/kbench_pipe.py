#!/usr/bin/env python3
"""Throughput with C independent contexts (streams) issuing steps round-robin (development tool)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np, torch
from wwhip.engine import Engine, frontend_params
from wwhip import _lib
model = sys.argv[1] if len(sys.argv) > 1 else "crnn"
clips = 256; steps = 400
rng = np.random.default_rng(0)
pcm = np.clip(rng.normal(0, 2000, (clips, 24000)), -32768, 32767).astype(np.int16)
if "zeros" in sys.argv: pcm[:] = 0  # DVFS probe: same instruction stream, no toggling
fp = frontend_params()
for C in (1, 2, 3, 4, 6, 8):
    ctxs = [_lib.Context(0) for _ in range(C)]
    engs = [Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models", "CRNN" if model == "crnn" else "Wavenet"), ctx=c) for c in ctxs]
    d = [torch.from_numpy(pcm).cuda() for _ in range(C)]
    out = [torch.zeros((clips, engs[0].n_out), device="cuda") for _ in range(C)]
    for k in range(3 * C):
        engs[k % C].clips_forward_dev(d[k % C].data_ptr(), clips, 24000, out[k % C].data_ptr(), fp)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(steps):
        engs[k % C].clips_forward_dev(d[k % C].data_ptr(), clips, 24000, out[k % C].data_ptr(), fp)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"C={C}: {el/steps*1e6:.1f} us/step, {steps*clips*150/el/1e6:.1f} M frames/s")
    for e in engs: e.close()
    for c in ctxs: c.close()
