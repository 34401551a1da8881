#!/usr/bin/env python3
"""Development tool: N steps of the clip path dealt round-robin to C contexts (the bench's pipelined job, nothing else),
for kernel traces.  usage: pipe_run.py [C] [steps]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np, torch
from wwhip.engine import Engine, frontend_params
from wwhip import _lib
C = int(sys.argv[1]) if len(sys.argv) > 1 else 4
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 400
clips = 256
rng = np.random.default_rng(0)
pcm = np.clip(rng.normal(0, 2000, (clips, 24000)), -32768, 32767).astype(np.int16)
d = [torch.roll(torch.from_numpy(pcm).cuda(), 37 * r, 1) for r in range(24)]
fp = frontend_params()
ctxs = [_lib.Context(0) for _ in range(C)]
engs = [Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/CRNN"), ctx=c) for c in ctxs]
out = [torch.zeros((clips, engs[0].n_out), device="cuda") for _ in range(C)]
for k in range(4 * C):
    engs[k % C].clips_forward_dev(d[k % 24].data_ptr(), clips, 24000, out[k % C].data_ptr(), fp)
torch.cuda.synchronize()
for rep in range(3):
    t0 = time.perf_counter()
    for k in range(steps):
        engs[k % C].clips_forward_dev(d[k % 24].data_ptr(), clips, 24000, out[k % C].data_ptr(), fp)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    print(f"C={C}: {el/steps*1e6:.1f} us/step", flush=True)
