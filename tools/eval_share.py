#!/usr/bin/env python3
"""Development (round 6): the reference evaluator's flow at hey-snips size - one GPU, and rank 0's share of a world of 8 run alone
(no communicator) - timed without kernel events: fastest and median of n passes, the fastest pass's host phases.
usage: eval_share.py [passes=9] [scale=1] [only8]"""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np, torch
from wwhip.evaluate import synth_testset_scaled, evaluate_reference_flow_sharded, SHARE_ONLY
from wwhip.models import engine_for
passes = int(sys.argv[1]) if len(sys.argv) > 1 else 9
scale = int(sys.argv[2]) if len(sys.argv) > 2 else 1
clips, labels = synth_testset_scaled(2529, 2529)
if scale > 1:
    clips = clips[:2529] * scale + clips[2529:] * scale
    labels = np.concatenate((np.ones(scale * 2529, np.uint8), np.zeros(scale * 2529, np.uint8)))
eng = engine_for(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/CRNN_softmax"), 0)
out = {}
only8 = "only8" in sys.argv[3:]
for world, comm in (((8, SHARE_ONLY),) if only8 else ((1, None), (8, SHARE_ONLY))):
    runs = []
    for i in range(passes + 2):
        torch.cuda.synchronize()
        tm = {"kernel_times": False}
        t0 = time.perf_counter()
        evaluate_reference_flow_sharded(eng, clips, labels, 0, world, comm, timing=tm)
        el = time.perf_counter() - t0
        if i >= 2:
            runs.append((el, tm))
    runs.sort(key=lambda x: x[0])
    out[f"world{world}"] = {"seconds_min": round(runs[0][0] * 1e3, 3), "seconds_median": round(runs[len(runs) // 2][0] * 1e3, 3),
                            "phases_ms": {k: round(v * 1e3, 3) for k, v in runs[0][1].items() if isinstance(v, float)}, "chunks": runs[0][1].get("chunks")}
if not only8:
    out["efficiency_8"] = round(out["world1"]["seconds_min"] / (8 * out["world8"]["seconds_min"]), 3)
print(json.dumps(out))
