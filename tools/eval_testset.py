#!/usr/bin/env python3
"""SURVEY 8(d) cfg 1 / cfg 4: evaluate the synthetic stand-in of the hey-snips test split
(2048 clips, 0.8-2.5 s, 10 % positives) - one-window accuracy + FRR/FA-per-hour sweep.

Single GPU:   python tools/eval_testset.py [--model CRNN_softmax] [--clips 2048] [--dump out.npz]
Sharded:      python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/eval_testset.py
              (utterances dealt longest-first round-robin, posteriors gathered over RCCL, rank 0 sweeps;
              fewer GPUs than ranks -> gloo, the ranks share the card)
--flow reference: the reference evaluator's own flow (wake-word clips through one never-reset Filter, the first num_wakewords
other clips joined into ONE stream, cut into contiguous posterior ranges over the ranks): evaluate_reference_flow_sharded.
The flows themselves are in wwhip.evaluate (tests/test_gpu_bench_eval.py runs them under -m gpu).
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]
import numpy as np


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="CRNN_softmax")
    ap.add_argument("--clips", type=int, default=2048)
    ap.add_argument("--backend", default=None)
    ap.add_argument("--flow", choices=["per_clip", "reference"], default="per_clip",
                    help="per_clip: every clip on its own (ring reset, 0.5 s of zeros each side); reference: utils/evaluate_models.py "
                         "main() - wake-word clips through one never-reset Filter, the first num_wakewords other clips joined into ONE "
                         "stream that is cut into contiguous posterior ranges over the ranks")
    ap.add_argument("--dump", default=None, help="rank 0 writes frr / fa_count / posteriors here (.npz)")
    args = ap.parse_args()
    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    dev = local % max(ndev, 1)
    torch.cuda.set_device(dev)
    dist, comm_dev = None, None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = args.backend or ("nccl" if ndev >= world else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=world)
        comm_dev = "cuda" if backend == "nccl" else "cpu"
    from wwhip.evaluate import synth_testset, evaluate_testset_sharded, evaluate_reference_flow_sharded
    from wwhip.models import engine_for
    eng = engine_for(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models", args.model), dev)
    clips, labels = synth_testset(args.clips)
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if args.flow == "reference":
        r = evaluate_reference_flow_sharded(eng, clips, labels, rank, world, comm_dev)
    else:
        r = evaluate_testset_sharded(eng, clips, labels, rank, world, comm_dev)
    if rank == 0:
        el = time.perf_counter() - t0
        line = {"model": args.model, "clips": len(clips), "world_size": world, "flow": args.flow, "seconds": el,
                "windows": r["windows"], "negative_hours": r["hours"], "frr_at_0.5_fa_per_hour": r["frr_at_0.5_fa_per_hour"],
                "fa_count_at_0.5": int(r["fa_count"][0]), "posterior_checksum": r["posterior_checksum"]}
        dump = {"frr": r["frr"], "fa_count": r["fa_count"], "positives": r["positives"], "negatives": r["negatives"],
                "checksum": r["posterior_checksum"]}
        if args.flow == "per_clip":
            line["audio_frames_per_s"] = sum((len(c) + 16000) // 160 for c in clips) / el
            line["one_window_accuracy"] = r["one_window_accuracy"]
            dump.update(sliding=r["sliding"], one=r["one_window_posteriors"])
        print(json.dumps(line))
        if args.dump:
            np.savez(args.dump, **dump)
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
