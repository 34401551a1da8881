#!/usr/bin/env python3
"""Sliding-window evaluation throughput (utils/evaluate_models.py flow, hop 2) on one long
negative stream: host mel in, posteriors out (PCIe inclusive) and the device-only kernel times."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np
from wwhip.engine import Engine, frontend_params
minutes = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
opts = dict(a.split("=", 1) for a in sys.argv[2:])  # e.g. crnn_tail_mfma=0 crnn_slide_min=0 (Engine.set_option); models=crnn
only = opts.pop("models", "")
rng = np.random.default_rng(0)
n = int(minutes * 60 * 16000)
pcm = np.clip(rng.normal(0, 2500, n), -32768, 32767).astype(np.int16)
out = {}
for name, prec in (("CRNN", "fp32"), ("CRNN", "bf16x3"), ("Wavenet", "fp32"), ("Wavenet", "bf16x3")):
    if only and (name.lower() != only or prec != "fp32"):
        continue
    eng = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models", name), precision=prec)
    for k, v in opts.items():
        if eng.is_crnn:
            eng.set_option(k, int(v))
    mel = eng.logmel([pcm])[0]
    eng.slide_forward(mel[:4000], 2)
    t0 = time.perf_counter(); mel = eng.logmel([pcm])[0]; t_fe = time.perf_counter() - t0
    t0 = time.perf_counter(); post = eng.slide_forward(mel, 2); t_sl = time.perf_counter() - t0
    eng.ctx.profile(True)
    eng.logmel([pcm]); eng.slide_forward(mel, 2)
    prof = eng.ctx.profile_read(); eng.ctx.profile(False)
    flop = {"CRNN": 8.04e6, "Wavenet": 20.66e6}[name] * len(post)
    kern_ms = {k: v["total_ms"] for k, v in prof.items()}
    model_ms = sum(v for k, v in kern_ms.items() if not k.startswith("logmel"))
    out[name if prec == "fp32" else name + "/" + prec] = {"posterior_checksum": float(np.sum(post, dtype=np.float64)), "audio_minutes": minutes, "mel_frames": int(len(mel)), "windows": int(len(post)),
                 "frontend_host_s": t_fe, "slide_host_s": t_sl,
                 "audio_frames_per_s_host": (n / 160) / (t_fe + t_sl),
                 "kernel_ms": kern_ms, "model_TFLOPs_device": flop / (model_ms * 1e-3) / 1e12,
                 "realtime_factor": minutes * 60 / (t_fe + t_sl)}
    eng.close()
print(json.dumps(out))
