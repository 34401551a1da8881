"""Front-end parity at scale (development tool): 256 ragged clips of mixed loudness, every log-mel frame against the C oracle."""
import os, sys
ROOT = "/root/repo" if os.path.isdir("/root/repo/tools") else os.environ.get("GRAFT_REPO_ROOT", ".")
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np
from wwhip.engine import Engine, frontend_params
from oracle.cpu import CpuOracle
eng = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/CRNN"))
ora = CpuOracle(eng.blob)
rng = np.random.default_rng(123)
pcm = []
for i in range(256):
    n = 24000 if i % 3 else int(rng.integers(600, 40000))
    amp = [30, 300, 3000, 20000][i % 4]
    x = rng.normal(0, amp, n)
    if i % 5 == 0: x += 8000 * np.sin(2 * np.pi * rng.uniform(100, 7000) * np.arange(n) / 16000)
    pcm.append(np.clip(x, -32768, 32767).astype(np.int16))
for precise in (True, False):
    got = eng.logmel(pcm, frontend_params(precise=precise))
    worst = 0.0
    for p, g in zip(pcm, got):
        w = ora.logmel(p)
        assert g.shape == w.shape
        if len(w): worst = max(worst, float(np.abs(g - w).max()))
    print("precise" if precise else "fast", "max |logmel - oracle| over", sum(len(g) for g in got), "frames:", worst)
