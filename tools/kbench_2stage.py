#!/usr/bin/env python3
"""Development: the clip path as a two-stage pipeline - ALL front-end launches on one HIP stream, ALL model launches on
another, events between them (front end of batch k + 1 beside the model of batch k) - against the bench's default of
four independent front-end -> model chains."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np, torch
from wwhip.engine import Engine, frontend_params
from wwhip import _lib
clips, steps, NB = 256, 2000, int(sys.argv[1]) if len(sys.argv) > 1 else 3
rng = np.random.default_rng(0)
pcm = np.clip(rng.normal(0, 2000, (clips, 24000)), -32768, 32767).astype(np.int16)
fp = frontend_params()
cA, cB = _lib.Context(0), _lib.Context(0)
eA = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/CRNN"), ctx=cA)
eB = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/CRNN"), ctx=cB)
import ctypes as C
hip = C.CDLL("libamdhip64.so")   # the runtime torch already loaded (same soname): raw events, ~1 us per call
def ev_new():
    e = C.c_void_p()
    assert hip.hipEventCreateWithFlags(C.byref(e), 2) == 0   # hipEventDisableTiming
    return e
sA, sB = C.c_void_p(cA.stream), C.c_void_p(cB.stream)
nf = (24000 - 512) // 160 + 1
d_pcm = torch.from_numpy(pcm).cuda()
so = torch.arange(clips + 1, dtype=torch.int64, device="cuda") * 24000
fo = torch.arange(clips + 1, dtype=torch.int64, device="cuda") * nf
mel = [torch.empty((clips * nf, 40), dtype=torch.float32, device="cuda") for _ in range(NB)]
rows = (torch.arange(clips, dtype=torch.int64, device="cuda") * nf)
valid = torch.full((clips,), nf, dtype=torch.int32, device="cuda")
out = torch.zeros((clips, eA.n_out), device="cuda")
fe_done = [ev_new() for _ in range(NB)]
md_done = [ev_new() for _ in range(NB)]
torch.cuda.synchronize()

def run(n):
    for k in range(n):
        b = k % NB
        if k >= NB:
            hip.hipStreamWaitEvent(sA, md_done[b], 0)          # the model of batch k - NB is done with this mel buffer
        eA.logmel_dev(d_pcm.data_ptr(), so.data_ptr(), fo.data_ptr(), clips, clips * nf, nf, mel[b].data_ptr(), fp)
        hip.hipEventRecord(fe_done[b], sA)
        hip.hipStreamWaitEvent(sB, fe_done[b], 0)
        eB.forward_windows_dev(mel[b].data_ptr(), clips * nf, rows.data_ptr(), valid.data_ptr(), clips, out.data_ptr())
        hip.hipEventRecord(md_done[b], sB)
    torch.cuda.synchronize()

run(50)
ref = out.clone()
t0 = time.perf_counter(); run(steps); el = time.perf_counter() - t0
print(f"two-stage, {NB} mel buffers: {el/steps*1e6:.1f} us/step, {steps*clips*150/el/1e6:.1f} M frames/s")
# reference: one chain
o2 = torch.zeros_like(out)
eA.clips_forward_dev(d_pcm.data_ptr(), clips, 24000, o2.data_ptr(), fp); cA.synchronize()
print("max |two-stage - clips_forward_dev|", float((o2 - ref).abs().max()))
