#!/usr/bin/env python3
"""Soak run (development tool): the full-size clip path N times per model/mode, every result compared bit for bit
with the first one - hunts ordering hazards that show up once in thousands of workgroups."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np, torch
from wwhip.engine import Engine, frontend_params
import bench
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
pcm = bench.synth_pcm(np.random.default_rng(5), 256)
d = torch.from_numpy(pcm).cuda()
for name, prec in (("CRNN", "fp32"), ("Wavenet", "fp32"), ("Wavenet", "bf16x3")):
    eng = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models", name), precision=prec)
    out = torch.zeros((256, eng.n_out), device="cuda")
    fp = frontend_params()
    eng.clips_forward_dev(d.data_ptr(), 256, 24000, out.data_ptr(), fp); eng.ctx.synchronize()
    ref = out.clone()
    bad = 0
    t0 = time.time()
    for i in range(N):
        out.zero_(); torch.cuda.synchronize()
        eng.clips_forward_dev(d.data_ptr(), 256, 24000, out.data_ptr(), fp); eng.ctx.synchronize()
        if not torch.equal(out, ref):
            bad += 1
    print(f"{name}/{prec}: {N} runs, {bad} mismatching, {time.time() - t0:.1f} s", flush=True)
    eng.close()
