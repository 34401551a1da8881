#!/usr/bin/env python3
"""Development: the reference evaluator's flow on the 2,048-clip stand-in, pass after pass, with host phases."""
import gc, os, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np, torch
from wwhip.evaluate import synth_testset, evaluate_reference_flow_sharded
from wwhip.models import engine_for
eng = engine_for(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/CRNN_softmax"), 0)
clips, labels = synth_testset(2048)
if "nogc" in sys.argv:
    gc.disable()
for i in range(12):
    tm = {}
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    r = evaluate_reference_flow_sharded(eng, clips, labels, timing=tm)
    el = time.perf_counter() - t0
    print(i, round(el * 1e3, 2), {k: round(v * 1e3, 2) for k, v in tm.items() if isinstance(v, float) and k != "device_ms"}, round(tm.get("device_ms", 0), 2), gc.get_count())
