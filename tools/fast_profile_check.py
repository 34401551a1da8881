#!/usr/bin/env python3
"""Evidence for the "fast profile" (fp32-FFT front end, optionally the split-bf16 CRNN contractions) at BASELINE cfg-1 scale:
does it leave the FA counts and the FRR array of the evaluation untouched?  Runs the 2,048-clip stand-in both ways
(per-clip flow: 118,643 windows; the reference's own flow: one joined negative stream) and prints one JSON object with
the posterior differences and, per flow, whether FA counts / FRR arrays are identical to the default profile (fp64 FFT,
fp32 MFMA).  tests/test_gpu_bench_eval.py asserts what this prints."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np
from wwhip.engine import Engine, frontend_params
from wwhip.evaluate import synth_testset, evaluate_testset_sharded, evaluate_reference_flow_sharded


def compare(eng, clips, labels):
    fast_fp = frontend_params(32767.0, True, 0.0, 160, False)
    out = {}
    base = evaluate_testset_sharded(eng, clips, labels)
    ref = evaluate_reference_flow_sharded(eng, clips, labels)
    for tag, precise, prec in (("fp32_fft", False, "fp32"), ("fp32_fft+bf16x3", False, "bf16x3"), ("fp64_fft+bf16x3", True, "bf16x3")):
        eng.set_precision(prec)
        a = evaluate_testset_sharded(eng, clips, labels, fp=None if precise else fast_fp)
        b = evaluate_reference_flow_sharded(eng, clips, labels, precise=precise)
        eng.set_precision("fp32")
        out[tag] = {
            "per_clip": {"max_abs_posterior_diff": float(np.abs(a["sliding"] - base["sliding"]).max()),
                         "fa_counts_identical": bool(np.array_equal(a["fa_count"], base["fa_count"])),
                         "fa_count_max_abs_diff": int(np.abs(a["fa_count"] - base["fa_count"]).max()),
                         "frr_identical": bool(np.array_equal(a["frr"], base["frr"])),
                         "frr_at_0.5": a["frr_at_0.5_fa_per_hour"], "frr_at_0.5_default": base["frr_at_0.5_fa_per_hour"]},
            "reference_flow": {"max_abs_posterior_diff": float(max(np.abs(b["negatives"] - ref["negatives"]).max(),
                                                                  np.abs(b["positives"] - ref["positives"]).max())),
                               "fa_counts_identical": bool(np.array_equal(b["fa_count"], ref["fa_count"])),
                               "fa_count_max_abs_diff": int(np.abs(b["fa_count"] - ref["fa_count"]).max()),
                               "frr_identical": bool(np.array_equal(b["frr"], ref["frr"])),
                               "frr_at_0.5": b["frr_at_0.5_fa_per_hour"], "frr_at_0.5_default": ref["frr_at_0.5_fa_per_hour"]}}
    return out


if __name__ == "__main__":
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    clips, labels = synth_testset(n)
    res = {}
    for name in ("CRNN_softmax", "CRNN"):
        eng = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models", name))
        res[name] = compare(eng, clips, labels)
        eng.close()
    print(json.dumps(res))
