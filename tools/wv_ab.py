#!/usr/bin/env python3
"""Development: the fp32 Wavenet's two block-loop forms side by side (HIP-event kernel times, 256 and 4,096 windows)."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np, torch
from wwhip.engine import Engine
from wwhip import _lib
ctx = _lib.Context(0)
eng = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/Wavenet"), ctx=ctx)
rng = np.random.default_rng(0)
out = {}
for n in (256, 4096):
    mel = torch.from_numpy(rng.uniform(0, 6.5, (n * 182, 40)).astype(np.float32)).cuda()
    row = torch.arange(n, dtype=torch.int64, device="cuda") * 182
    valid = torch.full((n,), 182, dtype=torch.int32, device="cuda")
    res = torch.zeros((n, 2), device="cuda")
    torch.cuda.synchronize()
    for form in (0, 1):
        eng.set_option("wavenet_rowmajor", form)
        for _ in range(5):
            eng.forward_windows_dev(mel.data_ptr(), n * 182, row.data_ptr(), valid.data_ptr(), n, res.data_ptr())
        ctx.synchronize()
        ctx.profile(True)
        for _ in range(30):
            eng.forward_windows_dev(mel.data_ptr(), n * 182, row.data_ptr(), valid.data_ptr(), n, res.data_ptr())
        p = ctx.profile_read(); ctx.profile(False)
        out[f"{n}_{'rowmajor' if form else 'transposed'}"] = {k: round(v["total_ms"] / v["calls"] * 1e3, 2) for k, v in p.items()}
        out[f"{n}_{'rowmajor' if form else 'transposed'}"]["checksum"] = float(res.double().sum().item())
print(json.dumps(out))
