"""Development tool: wall time of wwhip.evaluate.get_posterior (the reference evaluator's signature) over N synthetic wav files."""
import os, sys, time, wave, tempfile
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np
from wwhip.evaluate import get_posterior
N = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(0)
d = tempfile.mkdtemp()
files = []
for i in range(N):
    p = os.path.join(d, f"{i}.wav")
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(np.clip(rng.normal(0, 2000, int(rng.integers(12000, 40000))), -32768, 32767).astype(np.int16).tobytes())
    files.append(p)
mdir = os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/CRNN")
for et in ("false_negatives", "false_accepts"):
    get_posterior(mdir, "CRNN", et, files[:8], 20, 16000)
    t0 = time.perf_counter()
    out = get_posterior(mdir, "CRNN", et, files, 20, 16000)
    el = time.perf_counter() - t0
    print(et, N, "files", round(el * 1e3, 1), "ms", len(out), "values", round(el / N * 1e6, 1), "us/file")
