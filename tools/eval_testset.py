#!/usr/bin/env python3
"""SURVEY 8(d) cfg 1 / cfg 4: evaluate the synthetic stand-in of the hey-snips test split
(2048 clips, 0.8-2.5 s, 10 % positives) - one-window accuracy + FRR/FA-per-hour sweep.

Single GPU:   python tools/eval_testset.py [--model CRNN_softmax] [--clips 2048]
Sharded:      python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/eval_testset.py
              (utterances dealt longest-first round-robin, posteriors gathered over RCCL, rank 0 sweeps)
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
    args = ap.parse_args()
    import torch
    rank = int(os.environ.get("RANK", "0")); world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    ndev = torch.cuda.device_count()
    dev = local % max(ndev, 1)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        backend = args.backend or ("nccl" if ndev >= world else "gloo")
        dist.init_process_group(backend, rank=rank, world_size=world)
    from wwhip.evaluate import synth_testset, clip_posteriors, far_frr, frr_at_fa
    from wwhip.models import engine_for
    from wwhip import dist as D
    eng = engine_for(os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models", args.model), dev)
    clips, labels = synth_testset(args.clips)
    T = eng.window
    # global layout of the sliding posteriors (pure arithmetic, identical on every rank)
    n_frames = [((len(c) + 16000) - 512) // 160 + 1 for c in clips]
    n_win = np.array([max(0, (f - T) // 2 + 1) if f >= T else 0 for f in n_frames], np.int64)
    offs = np.concatenate(([0], np.cumsum(n_win)))
    mine = D.shard_by_length([len(c) for c in clips], world)[rank]
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    p_one, sliding = clip_posteriors(eng, [clips[i] for i in mine])
    slots = np.concatenate([np.arange(offs[i], offs[i + 1]) for i in mine]) if mine else np.zeros(0, np.int64)
    vals = np.concatenate(sliding) if sliding else np.zeros(0, np.float32)
    if world > 1:
        dev_t = "cuda" if dist.get_backend() == "nccl" else "cpu"
        all_slide = D.gather_posteriors(vals, slots, int(offs[-1]), device=dev_t)
        all_one = D.gather_posteriors(p_one, mine, len(clips), device=dev_t)
    else:
        all_slide = np.zeros(int(offs[-1]), np.float32); all_slide[slots] = vals
        all_one = np.zeros(len(clips), np.float32); all_one[mine] = p_one
    lab = labels.astype(bool)
    if rank == 0:
        # a clip shorter than the window yields no posterior (the reference's np.max would raise): 0
        pos = np.array([all_slide[offs[i]:offs[i + 1]].max() if offs[i + 1] > offs[i] else 0.0
                        for i in range(len(clips)) if lab[i]], np.float32)
        neg = np.concatenate([all_slide[offs[i]:offs[i + 1]] for i in range(len(clips)) if not lab[i]])
        hours = sum(len(clips[i]) + 16000 for i in range(len(clips)) if not lab[i]) / 16000 / 3600
        thr, frr, fa, cnt = far_frr(pos, neg, int(lab.sum()), hours, engine=eng)
        el = time.perf_counter() - t0
        audio_frames = sum((len(c) + 16000) // 160 for c in clips)
        print(json.dumps({"model": args.model, "clips": len(clips), "world_size": world, "seconds": el,
                          "audio_frames_per_s": audio_frames / el, "windows": int(offs[-1]),
                          "frr_at_0.5_fa_per_hour": frr_at_fa(frr, fa, 0.5), "fa_count_at_0.5": int(cnt[0]),
                          "one_window_accuracy": float(((all_one >= 0.5) == lab).mean()),
                          "posterior_checksum": float(all_slide.sum(dtype=np.float64))}))
    if dist is not None:
        dist.barrier(); dist.destroy_process_group()


if __name__ == "__main__":
    main()
