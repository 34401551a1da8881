#!/usr/bin/env python3
"""Development (round 6): gru_tail16_kernel alone - microseconds per launch of n explicit windows (front + matrix tail forced) under
whichever library WWHIP_LIB names; with -DGT16_LDS_PAD builds this is the kernel's time as a function of resident workgroups."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np, torch
from wwhip.engine import Engine
from wwhip import _lib
ctx = _lib.Context(0)
eng = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/CRNN"), ctx=ctx)
rng = np.random.default_rng(0)
sizes = [int(a) for a in sys.argv[1:]] or [4096, 8192, 16384, 32768]
nmax = max(sizes)
mel = torch.from_numpy(rng.uniform(0, 6.5, (nmax * 8 + 151, 40)).astype(np.float32)).cuda()
row = torch.arange(nmax, dtype=torch.int64, device="cuda") * 8
valid = torch.full((nmax,), 151, dtype=torch.int32, device="cuda")
res = torch.zeros((nmax, eng.n_out), device="cuda")
torch.cuda.synchronize()
out = {"lib": os.path.basename(os.environ.get("WWHIP_LIB", "libwwhip.so"))}
tail_opt = int(os.environ.get("TAIL_OPT", "2"))  # 2: gru_tail16_kernel, 3: gru_tail16h_kernel (hoisted projection, two groups per workgroup)
out["tail_opt"] = tail_opt
with eng.options(crnn_split_at=1, crnn_tail_mfma=tail_opt):
    for n in sizes:
        for _ in range(3):
            eng.forward_windows_dev(mel.data_ptr(), nmax * 8 + 151, row.data_ptr(), valid.data_ptr(), n, res.data_ptr())
        ctx.synchronize()
        ctx.profile(True)
        for _ in range(10):
            eng.forward_windows_dev(mel.data_ptr(), nmax * 8 + 151, row.data_ptr(), valid.data_ptr(), n, res.data_ptr())
        p = ctx.profile_read(); ctx.profile(False)
        out[n] = {k: round(v["total_ms"] / v["calls"] * 1e3, 1) for k, v in p.items()}
out["checksum"] = float(res[:sizes[-1]].double().sum().item())
import hashlib
out["sha"] = hashlib.sha256(res[:sizes[-1]].cpu().numpy().tobytes()).hexdigest()[:16]
print(json.dumps(out))
