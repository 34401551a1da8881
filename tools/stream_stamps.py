import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np
from wwhip.engine import Engine, StreamBank
S = 128
rng = np.random.default_rng(0)
eng = Engine(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models", "CRNN"))
bank = StreamBank(eng, S)
frames = np.clip(rng.normal(0, 2500, (64, S, 320)), -32768, 32767).astype(np.int16)
speech = np.ones(S, np.uint8)
for t in range(int(sys.argv[1]) if len(sys.argv) > 1 else 160): bank.step(frames[t % 64], speech)
