#!/usr/bin/env python3
"""BASELINE config 5 at the plugin surface by stream count (round 6): SpeechPipelineBank.step() over VadBank -> WakewordBank ->
ActivationTimeoutBank on a ContextBank (one library call per tick), raw VAD = speech on every stream (two posteriors per stream and
tick), p50 / p99 / p99.9 / max of the per-tick latency.  usage: pipeline_latency.py [ticks=20000] [model=CRNN] [S ...]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np
from wwhip.engine import Engine, StreamBank, frontend_params
from wwhip.pipeline import SpeechPipelineBank
from wwhip.vad import VadBank
from wwhip.wakeword import WakewordBank
from wwhip.activation_timeout import ActivationTimeoutBank
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 20000
name = sys.argv[2] if len(sys.argv) > 2 else "CRNN"
sizes = [int(a) for a in sys.argv[3:]] or [1, 16, 64, 128, 256, 512, 1024]
prec = "bf16x3" if name.lower().startswith("wavenet") else "fp32"
rng = np.random.default_rng(3)
out = {"model": name, "precision": prec, "ticks": ticks}
for S in sizes:
    eng = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models", name), precision=prec)
    frames = np.clip(rng.normal(0, 2500, (32, S, 320)), -32768, 32767).astype(np.int16)
    raw = np.ones(S, bool)

    class Source:
        frame = frames[0]
        def read(self): return self.frame
        def start(self): pass
        stop = close = start

    src = Source()
    wake = WakewordBank(S, posterior_threshold=0.5, bank=StreamBank(eng, S, frontend_params(32767.0, True, 0.0, 160, True)))
    pipe = SpeechPipelineBank(src, [VadBank(S, classifier=lambda f: raw), wake, ActivationTimeoutBank(S)], S)
    pipe.start()
    for t in range(200):
        src.frame = frames[t % 32]; pipe.step()
    lat = np.empty(ticks)
    for t in range(ticks):
        src.frame = frames[t % 32]
        t0 = time.perf_counter(); pipe.step(); lat[t] = time.perf_counter() - t0
    out[S] = {"p50_us": round(float(np.percentile(lat, 50)) * 1e6, 2), "p99_us": round(float(np.percentile(lat, 99)) * 1e6, 2),
              "p99.9_us": round(float(np.percentile(lat, 99.9)) * 1e6, 2), "max_us": round(float(lat.max()) * 1e6, 1),
              "share_of_a_20ms_tick": round(float(lat.mean()) / 0.020, 5), "fused": pipe._fused is not None}
    pipe.stop(); wake.close(); eng.close()
    print(S, out[S], flush=True)
print(json.dumps(out))
