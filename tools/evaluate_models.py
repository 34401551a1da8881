#!/usr/bin/env python3
"""The reference's evaluator command line (utils/evaluate_models.py:256-326) on the HIP path.

    python tools/evaluate_models.py --model_type CRNN --models_dir <dir with filter/encode/detect.tflite> \\
        --data_dir <hey-snips dir with test.json>/ --eval_dir data/evaluation/

Same arguments, same caches (<models_dir>/<model_type>_all_wakeword.pkl / _no_wakeword.pkl, the concatenated
negative wav under --eval_dir).  Instead of opening three matplotlib windows it prints the curves' summary as
JSON (add --plot for the windows when matplotlib is available).

Several GPUs: start it under `python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/evaluate_models.py ...`:
the wake-word files are dealt over the ranks, the ONE long negative wav is cut into N contiguous posterior ranges
(wwhip.evaluate.get_posterior_sharded), posteriors are gathered (RCCL; gloo when the ranks share a card), rank 0 sweeps."""
import argparse
import json
import os
import sys
from pathlib import Path

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, "wakeword-detection_amd")]

from wwhip import evaluate as E  # noqa: E402


def parse_args():
    p = argparse.ArgumentParser(description="Evaluates wakeword model(s), reports useful metrics.")
    p.add_argument("--model_type", type=str, default="CRNN", choices=["CRNN", "Wavenet"], help="Model type being evaluated.")
    p.add_argument("--models_dir", type=str, default=os.path.join(ROOT, "wakeword-detection_amd/assets/tf_lite_models/CRNN_softmax"),
                   help="Directory where trained models are stored.")
    p.add_argument("--data_dir", type=str, default="data/hey_snips_research_6k_en_train_eval_clean_ter/",
                   help="Directory with Hey Snips raw dataset")
    p.add_argument("--eval_dir", type=str, default="data/evaluation/", help="Directory to save and load concatenated wav files from")
    p.add_argument("--pos_samples", type=str, default="hey_snips_long.wav", help="File for concatenated positive class samples")
    p.add_argument("--neg_samples", type=str, default="not_hey_snips_long.wav", help="File for concatenated negative class samples")
    p.add_argument("--sample_rate", type=int, default=16000, help="Sample rate for audio (Hz)")
    p.add_argument("--frame_width", type=int, default=20, help="Frame width for audio in (ms)")
    p.add_argument("--examine_audio", default=False, action="store_true", help="Flag to examine problematic audio clips")
    p.add_argument("--plot", default=False, action="store_true", help="draw the three curves with matplotlib")
    args = p.parse_args()
    assert Path(args.models_dir).exists(), "Directory for TF-Lite models and results is not found!"
    return args


def main(args) -> int:
    rank, world, local = (int(os.environ.get(k, d)) for k, d in (("RANK", "0"), ("WORLD_SIZE", "1"), ("LOCAL_RANK", "0")))
    dist, comm_dev, dev = None, None, 0
    if world > 1:
        import torch
        import torch.distributed as dist
        ndev = torch.cuda.device_count()
        dev = local % max(ndev, 1)
        backend = "nccl" if ndev >= world else "gloo"  # fewer GPUs than ranks: the ranks share the card
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev)
        dist.init_process_group(backend, rank=rank, world_size=world)
        comm_dev = "cuda" if backend == "nccl" else "cpu"
    FAR_path = Path(os.path.join(args.eval_dir, args.neg_samples))
    wakeword_paths, not_wakeword_paths = E.testset_files(args.data_dir)
    num_wakewords = len(wakeword_paths)
    if rank == 0 and not FAR_path.exists():
        os.makedirs(args.eval_dir, exist_ok=True)
        E.concatenate_FA(not_wakeword_paths, num_wakewords, str(FAR_path), args.sample_rate)
    if dist is not None:
        dist.barrier()
    total_duration_hrs = E.duration_test(str(FAR_path), args.sample_rate) / 3600
    pos = E.load_posteriors(args.models_dir, args.model_type, args.frame_width, args.sample_rate, "false_negatives",
                            wakeword_paths, Path(os.path.join(args.models_dir, args.model_type + "_all_wakeword.pkl")),
                            args.examine_audio, rank, world, comm_dev, dev)
    neg = E.load_posteriors(args.models_dir, args.model_type, args.frame_width, args.sample_rate, "false_accepts",
                            [str(FAR_path)], Path(os.path.join(args.models_dir, args.model_type + "_no_wakeword.pkl")),
                            args.examine_audio, rank, world, comm_dev, dev)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank != 0:
        return 0
    res = E.plot_FRR_FAR(pos, neg, num_wakewords, total_duration_hrs, args.model_type, models_dir=args.models_dir, show=args.plot)
    print(json.dumps({"model_type": args.model_type, "num_wakewords": num_wakewords, "fa_hours": total_duration_hrs, "world_size": world,
                      "FA_count": [int(x) for x in res["FA_count"]],
                      "frr_at_0.5_fa_per_hour": res["frr_at_0.5_fa_per_hour"],
                      "FRR": [float(x) for x in res["FRR"]], "FA_per_hour": [float(x) for x in res["FAR"]],
                      "thresholds": [float(x) for x in res["thresholds"]]}))
    return 0


if __name__ == "__main__":
    sys.exit(main(parse_args()))
