#!/usr/bin/env python3
"""Development: the reference evaluator's flow at hey-snips scale (2,529 wake-word clips + the first 2,529 other clips joined
into a ~2 h stream) on one GPU, with the host / device split.  usage: eval_at_scale.py [n_wake] [passes]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np
from wwhip.evaluate import synth_testset_scaled, evaluate_reference_flow_sharded
from wwhip.models import engine_for
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2529
passes = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t0 = time.perf_counter()
clips, labels = synth_testset_scaled(n, n)
print(f"synth {time.perf_counter() - t0:.2f} s, {sum(len(c) for c in clips) / 16000 / 3600:.2f} h of audio", flush=True)
eng = engine_for(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/CRNN_softmax"), 0)
for i in range(passes):
    tm = {}
    t0 = time.perf_counter()
    r = evaluate_reference_flow_sharded(eng, clips, labels, timing=tm)
    el = time.perf_counter() - t0
    host = {k: round(v * 1e3, 2) for k, v in tm.items() if isinstance(v, float) and k != "device_ms"}
    print(json.dumps({"pass": i, "seconds": round(el, 4), "device_ms": round(tm["device_ms"], 2), "host_share": round(1 - tm["device_ms"] / 1e3 / el, 3),
                      "phases_ms": host, "kernels_ms": {k: round(v, 2) for k, v in tm["kernels_ms"].items()},
                      "windows": r["windows"], "hours": round(r["hours"], 3), "frr@0.5": r["frr_at_0.5_fa_per_hour"],
                      "fa0": int(r["fa_count"][0]), "checksum": r["posterior_checksum"]}), flush=True)
