#!/usr/bin/env python3
"""Development (round 6): a long run of the default tick (ONE launch, polled posteriors) against the two-launch / waited form on the
same inputs - 128 streams, random VAD runs, active stretches, single-stream and whole-bank resets - posteriors and counts compared
bit for bit every tick, the pipeline stages (VadBank -> WakewordBank -> ActivationTimeoutBank on a ContextBank) driven beside them on a
third bank.  usage: stream_soak.py [ticks=300000] [model=CRNN]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np
from wwhip.engine import Engine, StreamBank, frontend_params
from wwhip.context import ContextBank
from wwhip.vad import VadBank
from wwhip.wakeword import WakewordBank
from wwhip.activation_timeout import ActivationTimeoutBank
ticks = int(sys.argv[1]) if len(sys.argv) > 1 else 300000
name = sys.argv[2] if len(sys.argv) > 2 else "CRNN"
S = 128
rng = np.random.default_rng(11)
eng = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models", name))
fp = frontend_params(32767.0, True, 0.97, 160, True)
a, b = StreamBank(eng, S, fp), StreamBank(eng, S, fp, two_launch=True, sync_wait=True)
ctxs, vad, to = ContextBank(S), VadBank(S, vad_rise_delay=40, vad_fall_delay=60), ActivationTimeoutBank(S, min_active=100, max_active=600)
wake = WakewordBank(S, posterior_threshold=0.02, bank=StreamBank(eng, S, fp))
events = [0, 0]
ctxs.add_handler("activate", lambda c: events.__setitem__(0, events[0] + 1))
ctxs.add_handler("deactivate", lambda c: events.__setitem__(1, events[1] + 1))
frames = np.clip(rng.normal(0, 2500, (257, S, 320)), -32768, 32767).astype(np.int16)
speech = rng.random(S) < 0.6
active = np.zeros(S, bool)
n_post = n_reset = 0
t0 = time.perf_counter()
for t in range(ticks):
    flip = rng.random(S) < 0.04
    speech ^= flip
    active = np.where(rng.random(S) < 0.01, ~active, active)
    f = frames[t % 257]
    sp, ac = speech.astype(np.uint8), active.astype(np.uint8)
    (pa, na), (pb, nb) = a.step(f, sp, ac), b.step(f, sp, ac)
    if not (np.array_equal(na, nb) and np.array_equal(pa, pb)):
        print(json.dumps({"tick": t, "mismatch": True})); sys.exit(1)
    n_post += int(na.sum())
    vad(ctxs, f, raw=speech)
    wake.step(ctxs, f)
    to(ctxs, f)
    if rng.random() < 0.02:
        ids = np.flatnonzero(rng.random(S) < 0.05).astype(np.int32)
        if len(ids):
            a.reset(ids); b.reset(ids); n_reset += len(ids)
    if t % 50021 == 50020:
        a.reset(); b.reset(); n_reset += S
    if t % 20000 == 0:
        print(f"tick {t}: {n_post} posteriors identical so far, {events[0]} activations / {events[1]} deactivations, {time.perf_counter() - t0:.0f} s", flush=True)
print(json.dumps({"model": name, "ticks": ticks, "streams": S, "posteriors_compared": n_post, "stream_resets": n_reset, "identical": True,
                  "pipeline_activations": events[0], "pipeline_deactivations": events[1], "seconds": round(time.perf_counter() - t0, 1)}))
