#!/usr/bin/env python3
"""Development: the Wavenet kernel alone at several launch sizes (explicit windows), microseconds per launch by HIP events.
usage: wv_scale.py [bf16x3|fp32] [sizes...]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np, torch
from wwhip.engine import Engine
from wwhip import _lib
prec = sys.argv[1] if len(sys.argv) > 1 else "bf16x3"
sizes = [int(a) for a in sys.argv[2:]] or [256, 512, 1024, 4096, 16384]
ctx = _lib.Context(0)
eng = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/Wavenet"), ctx=ctx, precision=prec)
nmax = max(sizes)
rng = np.random.default_rng(0)
mel = torch.from_numpy(rng.uniform(0, 6.5, (nmax * 2 + 200, 40)).astype(np.float32)).cuda()
row = torch.arange(nmax, dtype=torch.int64, device="cuda") * 2
valid = torch.full((nmax,), eng.window, dtype=torch.int32, device="cuda")
res = torch.zeros((nmax, eng.n_out), device="cuda")
torch.cuda.synchronize()
out = {}
for n in sizes:
    for _ in range(3):
        eng.forward_windows_dev(mel.data_ptr(), mel.shape[0], row.data_ptr(), valid.data_ptr(), n, res.data_ptr())
    ctx.synchronize()
    ctx.profile(True)
    for _ in range(10):
        eng.forward_windows_dev(mel.data_ptr(), mel.shape[0], row.data_ptr(), valid.data_ptr(), n, res.data_ptr())
    p = ctx.profile_read(); ctx.profile(False)
    us = sum(v["total_ms"] / v["calls"] for v in p.values()) * 1e3
    out[n] = round(us, 1)
print(prec, out, "checksum", float(res[:256].sum()))
