#!/usr/bin/env python3
"""Development: explicit-window CRNN launches of n windows in the three forms - one fused kernel (crnn_split_at=0), front + gru_tail16_kernel,
front + gru_tail_kernel - model-kernel time per launch (HIP events), to place WW_OPT_CRNN_SPLIT_AT and the tail choice."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np, torch
from wwhip.engine import Engine
from wwhip import _lib
ctx = _lib.Context(0)
eng = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/CRNN"), ctx=ctx)
rng = np.random.default_rng(0)
sizes = [int(a) for a in sys.argv[1:]] or [512, 1024, 1536, 2048, 3072, 4096, 6144, 8192, 16384]
nmax = max(sizes)
mel = torch.from_numpy(rng.uniform(0, 6.5, (nmax * 151, 40)).astype(np.float32)).cuda()
row = torch.arange(nmax, dtype=torch.int64, device="cuda") * 151
valid = torch.full((nmax,), 151, dtype=torch.int32, device="cuda")
res = torch.zeros((nmax, eng.n_out), device="cuda")
torch.cuda.synchronize()
out = {}
for n in sizes:
    rec = {}
    for tag, opts in (("fused", dict(crnn_split_at=0)), ("front+tail16", dict(crnn_split_at=1, crnn_tail_mfma=2)),
                      ("front+tail_valu", dict(crnn_split_at=1, crnn_tail_mfma=0))):
        with eng.options(**opts):
            for _ in range(3):
                eng.forward_windows_dev(mel.data_ptr(), nmax * 151, row.data_ptr(), valid.data_ptr(), n, res.data_ptr())
            ctx.synchronize()
            ctx.profile(True)
            for _ in range(20):
                eng.forward_windows_dev(mel.data_ptr(), nmax * 151, row.data_ptr(), valid.data_ptr(), n, res.data_ptr())
            p = ctx.profile_read(); ctx.profile(False)
        rec[tag] = round(sum(v["total_ms"] / v["calls"] for v in p.values()) * 1e3, 1)
    out[n] = rec
    print(n, rec, flush=True)
