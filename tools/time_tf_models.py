#!/usr/bin/env python3
"""utils/time_tf_models.py on the HIP path: mean batch-1 latency of encode (+ detect) through the TFLiteModel surface.

    python tools/time_tf_models.py --model_type CRNN --tf_lite_model_dir wakeword-detection_amd/assets/tf_lite_models/CRNN

The reference times the Keras model against the TF-Lite interpreters (``:14-70``); Keras is out of scope here, the TF-Lite half
is what ``TFLiteModel`` replaces.  The reference's loop never invokes the detector (``:63-66``: it sets the detector's input and
stops - SURVEY quirk C7); this script reports that figure (``encode_only``) AND the complete encode + detect call, and next to
them the same window through ``Engine.forward`` (one C call for encode + detect).  ``--time_quantized`` times the float16-weight
variant (``wwhip.weights.quantize_fp16``: what TFLite's float16 quantisation stores; the reference's ``*-quant.tflite`` files are
not shipped)."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]

import numpy as np  # noqa: E402


def mean_seconds(fn, num_runs):
    fn()  # prime: the first call sizes workspaces (the reference primes its interpreters as well, :54-58)
    total = 0.0
    for _ in range(num_runs):
        start = time.perf_counter()
        fn()
        total += time.perf_counter() - start
    return total / num_runs


def time_models(args):
    from spokestack.models.tensorflow import TFLiteModel  # the reference's import path (wwhip.models)
    from wwhip.engine import Engine
    encode = TFLiteModel(model_path=os.path.join(args.tf_lite_model_dir, "encode.tflite"))
    detect = TFLiteModel(model_path=os.path.join(args.tf_lite_model_dir, "detect.tflite"))
    rng = np.random.default_rng(0)
    if args.model_type == "CRNN":
        x = rng.random((1, args.num_features, 151 if args.timesteps == 182 else args.timesteps, 1)).astype(np.float32)  # [1, 40, 151, 1]
    else:
        x = rng.random((1, args.timesteps, args.num_features)).astype(np.float32)                                      # [1, 182, 40]
    out = {"model_type": args.model_type, "num_runs": args.num_runs}
    print(f"Running timings on the HIP-backed {args.model_type} models (TFLiteModel surface)")
    out["encode_only"] = mean_seconds(lambda: encode(x), args.num_runs)

    def both():
        enc = np.array(encode(x))
        return detect(enc.squeeze(0) if args.model_type == "CRNN" else enc[0])
    out["encode_and_detect"] = mean_seconds(both, args.num_runs)
    eng = Engine(args.tf_lite_model_dir, weights_fp16=args.time_quantized)
    win = rng.random((1, eng.window, eng.n_mel)).astype(np.float32)
    out["engine_forward"] = mean_seconds(lambda: eng.forward(win), args.num_runs)
    eng.close()
    print(f"TF-Lite surface, encode only (what the reference's loop times): {out['encode_only']} secs")
    print(f"TF-Lite surface, encode + detect: {out['encode_and_detect']} secs")
    print(f"one C call for encode + detect{' (float16 weights)' if args.time_quantized else ''}: {out['engine_forward']} secs")
    print(json.dumps(out))
    return out


def parse_args():
    p = argparse.ArgumentParser(description="Timing script for the wake-word models (batch 1).")
    p.add_argument("--model_type", type=str, default="Wavenet", choices=["CRNN", "Wavenet"], help="Model type being evaluated.")
    p.add_argument("--tf_lite_model_dir", type=str, default="", help="Directory with the .tflite models (default: the shipped ones)")
    p.add_argument("--num_features", type=int, default=40, help="Number of features per-timestep")
    p.add_argument("--timesteps", type=int, default=182, help="Number of timesteps per example")
    p.add_argument("--num_runs", type=int, default=10, help="Number of runs to get average inference time")
    p.add_argument("--time_quantized", action="store_true", help="Time the float16-weight variant")
    a = p.parse_args()
    if not a.tf_lite_model_dir:
        a.tf_lite_model_dir = os.path.join(ROOT, "wakeword-detection_amd", "assets", "tf_lite_models", a.model_type)
    return a


if __name__ == "__main__":
    start = time.time()
    time_models(parse_args())
    print(f"Script completed in {time.time() - start:.2f} secs")
