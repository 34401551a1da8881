#!/usr/bin/env python3
"""utils/time_tf_models.py on the HIP path: mean batch-1 latency of encode + detect on a random window (host array
in, host posterior out), for the float32 model, optionally its float16-weight variant (--time_quantized) and, for
the Wavenet, the split-bf16 mode.  The reference times Keras `model.predict` and two TFLite interpreters the same
way (10 runs after one priming call); neither TensorFlow nor the `-quant.tflite` files exist here, so only the
TF-Lite column has a counterpart."""
import argparse
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]

import numpy as np  # noqa: E402

from wwhip.engine import Engine  # noqa: E402


def time_models(eng: Engine, num_runs: int) -> float:
    X = np.array(np.random.random_sample((1, eng.window, eng.n_mel)), dtype=np.float32)
    eng.forward(X)  # prime (allocations, first launch)
    total = 0.0
    for _ in range(num_runs):
        start = time.perf_counter()
        eng.forward(X)
        total += time.perf_counter() - start
    return total / num_runs


def parse_args():
    p = argparse.ArgumentParser(description="Gets inference timings for the HIP versions of the wake-word models.")
    p.add_argument("--model_type", type=str, default="Wavenet", choices=["CRNN", "Wavenet"], help="Model type being evaluated.")
    p.add_argument("--tf_lite_model_dir", type=str, default=None, help="Directory with filter/encode/detect.tflite")
    p.add_argument("--num_runs", type=int, default=10, help="Number of runs to get average inference time")
    p.add_argument("--time_quantized", action="store_true", help="Time the float16-weight version of the models")
    return p.parse_args()


def main(args):
    start = time.time()
    mdir = args.tf_lite_model_dir or os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models", args.model_type)
    modes = [("float32", dict())]
    if args.time_quantized:
        modes.append(("float16 weights", dict(weights_fp16=True)))
    if args.model_type == "Wavenet":
        modes.append(("split-bf16 MFMA", dict(precision="bf16x3")))
    for label, kw in modes:
        eng = Engine(mdir, **kw)
        if eng.is_crnn != (args.model_type == "CRNN"):
            raise ValueError(f"{mdir} does not hold a {args.model_type} model")
        print(f"HIP {args.model_type} ({label}) average time: {time_models(eng, args.num_runs)} secs")
        eng.close()
    print(f"Script completed in {time.time() - start:.2f} secs")
    return 0


if __name__ == "__main__":
    sys.exit(main(parse_args()))
