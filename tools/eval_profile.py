#!/usr/bin/env python3
"""Development: cProfile of the reference evaluator's flow at hey-snips size on one GPU - world 1, or one rank's share of a
world of 8 run alone (no communicator).  usage: eval_profile.py [world=8]"""
import cProfile, os, pstats, sys, time
ROOT = os.environ.get("GRAFT_REPO_ROOT") or os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
from wwhip.evaluate import synth_testset_scaled, evaluate_reference_flow_sharded, SHARE_ONLY
from wwhip.models import engine_for
world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
clips, labels = synth_testset_scaled(2529, 2529)
eng = engine_for(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/CRNN_softmax"), 0)
comm = SHARE_ONLY if world > 1 else None
for _ in range(3):
    evaluate_reference_flow_sharded(eng, clips, labels, 0, world, comm)
t0 = time.perf_counter()
evaluate_reference_flow_sharded(eng, clips, labels, 0, world, comm)
print("unprofiled pass: %.3f ms" % ((time.perf_counter() - t0) * 1e3))
pr = cProfile.Profile()
pr.enable()
evaluate_reference_flow_sharded(eng, clips, labels, 0, world, comm)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(28)
