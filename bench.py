#!/usr/bin/env python3
"""Headline benchmark: audio frames/s of the wake-word hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--model crnn|wavenet] [--clips 256]

One step = one pass of the whole hot path over one batch of synthetic 16 kHz clips that is
already resident in HBM: int16 PCM [clips, 24000] -> log-mel -> one zero-padded window per
clip -> encode + detect -> posteriors [clips, n_out]   (BASELINE.json configs[1] for CRNN,
configs[2] for Wavenet; 1 audio frame = one 10 ms hop = 160 samples, 150 per 1.5 s clip).

N > 1: launched by torch.distributed.run, one rank per GPU; every rank owns its own shard of
clips (weak scaling, no data-path collective).  The only exchange is the posterior gather
(RCCL all_gather of the K*clips*n_out floats each rank produced), done once at the end of the
timed region, as the offline evaluator would do after its shard is finished.

Rank 0 prints ONE JSON line (schema in the task contract) including
  roofline     - dominant kernel, measured with HIP events around every launch in a separate
                 pass of the same K steps (ww_profile_enable), priced against the fp32 MFMA
                 peak (157.3 TFLOP/s) or HBM (8 TB/s) - figures in DESIGN.md section 4
  cpu_baseline - the C restatement in oracle/ (NOT TFLite) timed on this box's host cores on
                 a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.join(ROOT, "wakeword-detection_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)

import numpy as np  # noqa: E402

SAMPLES = 24000          # 1.5 s @ 16 kHz
FRAMES_PER_CLIP = 150    # audio frames (10 ms hops) per clip
PEAK_F32_MFMA = 157.3e12  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2, 256 CUs
PEAK_HBM = 8.0e12
PEAK_BF16_MFMA = 2.5e15   # dense bf16, MI355X_MICROARCH.md
PEAK_F64_VALU = 78.6e12   # fp64 vector FMA peak


def synth_pcm(rng, n_clips):
    """Gaussian noise sigma=2000 LSB + linear chirp 200->4000 Hz at 8000 LSB."""
    t = np.arange(SAMPLES) / 16000.0
    phase = 2 * np.pi * (200.0 * t + 0.5 * (4000.0 - 200.0) / 1.5 * t * t)
    x = rng.normal(0.0, 2000.0, (n_clips, SAMPLES)) + 8000.0 * np.sin(phase)[None, :]
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def kernel_flops(eng, n_clips):
    """Algorithmic FLOPs (2*MAC) and HBM bytes per launch of each kernel for one step."""
    nf = (SAMPLES - 512) // 160 + 1
    out = {}
    fe_bytes = n_clips * (SAMPLES * 2 + nf * 40 * 4)
    fe_flops = n_clips * nf * 13.9e3
    out["logmel_kernel<f64>"] = ("hbm", fe_bytes, fe_flops)
    out["logmel_kernel<f32>"] = ("hbm", fe_bytes, fe_flops)
    if eng.is_crnn:
        c = eng.bundle.crnn
        M = c.out_t * c.out_f
        K = c.conv_w.shape[1] * c.conv_w.shape[2]
        H = c.units
        out["conv5x20_kernel"] = ("mfma", n_clips * (eng.window * 40 * 4 + M * 32 * 4), n_clips * 2.0 * M * K * 32)
        out["gemm_nt_kernel<gru1>"] = ("mfma", n_clips * c.out_t * (M // c.out_t * 32 + 6 * H) * 4 + 6 * H * c.out_f * 32 * 4,
                                       n_clips * 2.0 * c.out_t * (c.out_f * 32) * 6 * H)
        out["gemm_nt_kernel<gru2>"] = ("mfma", n_clips * c.out_t * (2 * H + 6 * H) * 4, n_clips * 2.0 * c.out_t * 2 * H * 6 * H)
        out["gru_kernel<seq>"] = ("latency", n_clips * c.out_t * (6 * H + 2 * H) * 4, n_clips * 2.0 * c.out_t * 2 * 3 * H * H)
        out["gru_kernel<last+head>"] = ("latency", n_clips * c.out_t * 6 * H * 4,
                                        n_clips * (2.0 * c.out_t * 2 * 3 * H * H + 2 * (64 * 64 + 64 * c.n_out)))
    else:
        w = eng.bundle.wavenet
        macs = w.n_frames * (w.n_mel * w.channels) + sum(
            w.n_frames * (3 * w.channels * 2 * w.channels + w.channels * ((w.channels if b.w_res is not None else 0) + w.skip_channels))
            for b in w.blocks) + w.n_frames * (w.skip_channels * w.skip_channels + w.skip_channels * eng.n_out)
        out["wavenet_kernel"] = ("mfma", n_clips * (eng.window * 40 * 4 + eng.n_out * 4), n_clips * 2.0 * macs)
        out["wavenet_kernel<bf16x3>"] = out["wavenet_kernel"]
    return out


def cpu_baseline(eng, pcm_sample, budget_s=10.0):
    """Time the C restatement (oracle/ww_oracle.c, all host threads) on a bounded sample."""
    from oracle import cpu as ocpu
    ora = ocpu.CpuOracle(eng.blob)
    # a 1-GPU box grants this job 16 CPUs (of many more logical ones): use exactly that share
    threads = ocpu.set_threads(max(1, min(16, os.cpu_count() or 1)))

    def one_pass(clips):
        wins = np.zeros((len(clips), eng.window, 40), np.float32)
        for i, c in enumerate(clips):
            mel = ora.logmel(c)
            n = min(len(mel), eng.window)
            wins[i, :n] = mel[:n]
        return ora.forward(wins)

    one_pass(pcm_sample[:threads])  # warm up (thread pool, tables)
    t0 = time.perf_counter()
    done = 0
    n = len(pcm_sample)
    while True:
        one_pass(pcm_sample)
        done += n
        el = time.perf_counter() - t0
        if el > budget_s:
            break
    return {
        "value": done * FRAMES_PER_CLIP / el,
        "unit": "audio frames/s",
        "cores": threads,
        "kind": "port",
        "sample": f"{done} clips x 1.5 s ({n}-clip sample of the same synthetic batch, repeated), "
                  f"{el:.1f} s wall; C restatement oracle/ww_oracle.c with OpenMP over clips, NOT TFLite",
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--model", choices=["crnn", "wavenet"], default="crnn")
    ap.add_argument("--clips", type=int, default=256)
    ap.add_argument("--rotate", type=int, default=24, help="distinct resident input batches to rotate over")
    ap.add_argument("--pipeline", type=int, default=4,
                    help="independent contexts (HIP streams) the steps are dealt to round-robin; batches are "
                         "independent, so consecutive steps may overlap on the GPU")
    ap.add_argument("--fast-frontend", action="store_true", help="fp32 FFT instead of the reference's fp64")
    ap.add_argument("--precision", choices=["auto", "fp32", "bf16x3", "bf16x6"], default="auto",
                    help="model contractions: fp32 MFMA, or (Wavenet only) three bf16 MFMAs on split operands with "
                         "fp32 accumulate; auto = fp32 for CRNN (BASELINE cfg 2), bf16x3 for Wavenet (cfg 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} needs WORLD_SIZE={args.gpus} (launch with torch.distributed.run)")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; no CPU fallback exists for the hot path")
    # WW_BENCH_BACKEND=gloo is a rehearsal mode for a one-GPU box (ranks share GPU 0, the gather
    # goes through host memory); the driver's multi-GPU runs use RCCL ("nccl"), one GPU per rank.
    backend = os.environ.get("WW_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    comm_dev = "cuda" if backend == "nccl" else "cpu"

    from wwhip.engine import Engine, frontend_params
    from wwhip import _lib
    P = max(1, args.pipeline)
    ctxs = [_lib.Context(local_rank) for _ in range(P)]
    model_dir = os.path.join(PKG, "assets", "tf_lite_models", "CRNN" if args.model == "crnn" else "Wavenet")
    precision = args.precision if args.precision != "auto" else ("fp32" if args.model == "crnn" else "bf16x3")
    if args.model == "crnn" and precision != "bf16x6":
        precision = "fp32"
    engs = [Engine(model_dir, device=local_rank, ctx=c, precision=precision) for c in ctxs]
    ctx, eng = ctxs[0], engs[0]
    fp = frontend_params(32767.0, True, 0.0, 160, not args.fast_frontend)

    # R distinct resident input batches, rotated step by step: R * 12.3 MB exceeds the 256 MiB
    # Infinity Cache, so every step's PCM comes from HBM rather than from a cache-warm copy.
    rng = np.random.default_rng(1000 + rank)
    R = max(1, min(args.rotate, args.steps))
    pcm0 = synth_pcm(rng, args.clips)
    d_pcm = []
    for r in range(R):
        if r == 0:
            d_pcm.append(torch.from_numpy(pcm0).cuda())
        else:  # cheap decorrelated variants: roll + sign flip, generated on the device
            d_pcm.append(torch.roll(d_pcm[0], shifts=37 * r, dims=1) * (1 if r % 2 == 0 else -1))
    K = args.steps
    d_outs = [torch.zeros((args.clips, eng.n_out), dtype=torch.float32, device="cuda") for _ in range(R)]
    d_all = torch.zeros((K, args.clips, eng.n_out), dtype=torch.float32, device="cuda")
    slot_of_step = torch.arange(K, device="cuda") % R
    torch.cuda.synchronize()

    def step(k, only0=False):
        r = k % R
        e = eng if only0 else engs[k % P]
        e.clips_forward_dev(d_pcm[r].data_ptr(), args.clips, SAMPLES, d_outs[r].data_ptr(), fp)

    def sync_all():
        for c in ctxs:
            c.synchronize()

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for k in range(max(args.warmup, R * P)):  # every (context, batch) pair seen once: graphs captured
        step(k)
    sync_all()
    if dist is not None:  # warm up the communicator outside the timed region
        src = d_all.to(comm_dev)
        tmp = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(tmp, src)
        del tmp, src
    barrier()
    t0 = time.perf_counter()
    for k in range(K):
        step(k)
    sync_all()
    if dist is not None:
        # posterior gather, once per job: every rank contributes the K*clips*n_out floats it produced
        d_all = torch.stack(d_outs)[slot_of_step]  # two device kernels, not K copies
        src = d_all.to(comm_dev)
        gathered = [torch.empty_like(src) for _ in range(world)]
        dist.all_gather(gathered, src)
    barrier()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=comm_dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    # ---- same region once more with the fp32-FFT front end (precise=0), reported as a side figure:
    # posteriors stay within 5e-6 of the fp64-FFT path (tools/f32_error.py), log-mel within 1.4e-4
    alt = None
    if not args.fast_frontend:
        fp_fast = frontend_params(32767.0, True, 0.0, 160, False)

        def step_fast(k):
            engs[k % P].clips_forward_dev(d_pcm[k % R].data_ptr(), args.clips, SAMPLES, d_outs[k % R].data_ptr(), fp_fast)
        for k in range(R * P):
            step_fast(k)
        sync_all()
        barrier()
        t1 = time.perf_counter()
        for k in range(K):
            step_fast(k)
        sync_all()
        barrier()
        alt = time.perf_counter() - t1
        for k in range(R * P):  # restore the fp64-front-end outputs
            step(k)
        sync_all()

    # ---- Wavenet: the same region with fp32 MFMA contractions (parity mode), reported alongside (cfg 3)
    alt_fp32 = None
    if precision == "bf16x3":
        for e in engs:
            e.set_precision("fp32")
        for k in range(R * P):
            step(k)
        sync_all()
        barrier()
        t2 = time.perf_counter()
        for k in range(K):
            step(k)
        sync_all()
        barrier()
        alt_fp32 = time.perf_counter() - t2
        posts_fp32 = torch.stack(d_outs).cpu().numpy()
        for e in engs:
            e.set_precision("bf16x3")
        for k in range(R * P):
            step(k)
        sync_all()
        alt_fp32_maxdiff = float(np.abs(torch.stack(d_outs).cpu().numpy() - posts_fp32).max())

    # ---- per-kernel pass (HIP events around every launch), same K steps
    ctx.profile(True)
    for k in range(K):
        step(k, only0=True)
    prof = ctx.profile_read()
    ctx.profile(False)
    posts = d_outs[0].cpu().numpy()

    if rank == 0:
        total_frames = world * K * args.clips * FRAMES_PER_CLIP
        kf = kernel_flops(eng, args.clips)
        per_kernel = {}
        dom, dom_ms = None, -1.0
        for name, rec in prof.items():
            avg = rec["total_ms"] / max(rec["calls"], 1)
            per_kernel[name] = round(avg * 1e3, 3)  # microseconds
            if name in kf and avg > dom_ms:
                dom, dom_ms = name, avg
        bound, nbytes, flops = kf[dom]
        if bound == "hbm":
            roof = {"bound": "hbm", "achieved": nbytes / (dom_ms * 1e-3) / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s"}
        elif dom.endswith("<bf16x3>"):
            # three bf16 MFMAs per product: priced against the dense bf16 peak with 3x the algorithmic FLOPs
            roof = {"bound": "mfma", "achieved": 3.0 * flops / (dom_ms * 1e-3) / 1e12, "peak": PEAK_BF16_MFMA / 1e12,
                    "unit": "TFLOP/s", "note": "split-bf16: 3 bf16 MFMA products per algorithmic product; "
                    f"algorithmic rate {flops / (dom_ms * 1e-3) / 1e12:.1f} TFLOP/s"}
        else:
            roof = {"bound": "mfma", "achieved": flops / (dom_ms * 1e-3) / 1e12, "peak": PEAK_F32_MFMA / 1e12, "unit": "TFLOP/s"}
        roof["frac"] = roof["achieved"] / roof["peak"]
        if bound == "hbm" and not args.fast_frontend:
            # the front end is priced against HBM as SURVEY 8(d) defines it; what actually bounds it is the fp64
            # vector ALU (Hann product, two radix-16 DFTs and the untangling are all float64 like the reference)
            tf64 = flops / (dom_ms * 1e-3) / 1e12
            roof["fp64_valu"] = {"achieved_TFLOPs": tf64, "peak_TFLOPs": PEAK_F64_VALU / 1e12, "frac": tf64 * 1e12 / PEAK_F64_VALU,
                                 "note": "13.9 kFLOP per frame (SURVEY 8d); mostly adds, so half of the FMA peak is the ceiling"}
        roof["traffic"] = None
        try:  # HBM-side bytes per launch from the committed PMC passes of the same workload
            pmc = json.load(open(os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json")))["kernels"].get(dom)
            if pmc and args.clips == 256:
                roof["traffic"] = (2 * pmc["FETCH_SIZE_KiB"] + pmc["WRITE_SIZE_KiB"]) * 1024
                roof["traffic_note"] = ("2 x FETCH_SIZE + WRITE_SIZE (gfx950 reports half of 16-byte-per-lane streaming reads: "
                                        "MI355X_MICROARCH.md), rocprofv3 --pmc in separate passes, bytes per launch "
                                        "(profiles/r01/pmc_traffic.json)")
                roof["algorithmic_bytes"] = nbytes
            sq = json.load(open(os.path.join(ROOT, "profiles", "r01", "pmc_traffic.json"))).get("sq_counters_logmel_f64_v6_final")
            if sq and dom == "logmel_kernel<f64>" and args.clips == 256:
                # what bounds the kernel in practice, from the committed SQ counter passes (per CU: 4 vector ALUs, 1 LDS pipe;
                # SQ_ACTIVE_INST_* count quad-cycles, SQ_BUSY_CYCLES is summed over the 32 shader engines)
                cyc = sq["SQ_BUSY_CYCLES"] / 32.0
                roof["pipes_busy"] = {"valu": round(sq["SQ_ACTIVE_INST_VALU"] * 4 / 1024.0 / cyc, 3),
                                      "lds": round(sq["SQ_LDS_IDX_ACTIVE"] / 256.0 / cyc, 3),
                                      "lds_bank_conflict_share": round(sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"], 3),
                                      "wave_cycles_in_waitcnt": round(sq["SQ_WAIT_ANY"] / sq["SQ_WAVE_CYCLES"], 3),
                                      "source": "profiles/r01/pmc_traffic.json: sq_counters_logmel_f64_v6_final (rocprofv3 --pmc, tools/pmc_fe.sh)"}
        except (OSError, KeyError, ValueError):
            pass
        roof["kernel"] = dom
        roof["kernel_avg_us"] = round(dom_ms * 1e3, 3)
        roof["all_kernels_avg_us"] = per_kernel
        roof["method"] = "HIP events around each launch (ww_profile_enable), K steps right after the timed region"
        line = {
            "metric": "audio frames/sec (16 kHz, 40-mel, 10 ms hop)",
            "value": total_frames / elapsed,
            "unit": "audio frames/s",
            "n_gpus": world,
            "steps": K,
            "warmup": args.warmup,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32" if precision == "fp32" else "bf16x3 (split-bf16 products, f32 accumulate)",
            "data": "synthetic",
            "config": {
                "workload": (f"{'CRNN' if args.model == 'crnn' else 'Wavenet'} forward, batch={args.clips}x1.5 s synthetic "
                             f"16 kHz clips per GPU, PCM resident in HBM -> log-mel ({'fp32' if args.fast_frontend else 'fp64'} FFT) "
                             f"-> {eng.window}x40 window -> encode+detect, {'fp32 MFMA' if precision == 'fp32' else 'split-bf16 MFMA (bf16x3), fp32 accumulate'}"),
                "clips_per_gpu": args.clips,
                "samples_per_clip": SAMPLES,
                "resident_input_batches_rotated": R,
                "pipelined_contexts": P,
                "weights": "reference tf_lite_models (shipped fp32 weights)",
                "parallelism": f"utterance-sharded x{world}, posterior all_gather once per job" if world > 1 else "single GPU",
            },
            "roofline": roof,
            "posterior_checksum": float(np.sum(posts, dtype=np.float64)),
            "alt_fp32_fft_frontend": None if alt is None else {
                "value": total_frames / alt, "unit": "audio frames/s", "ms_per_step": alt / K * 1e3,
                "note": "same job with ww_frontend_params.precise=0 (fp32 butterflies); per-rank time, not max-reduced"},
        }
        if alt_fp32 is not None:
            line["alt_fp32_mfma_parity_mode"] = {
                "value": total_frames / alt_fp32, "unit": "audio frames/s", "ms_per_step": alt_fp32 / K * 1e3,
                "max_abs_posterior_diff_vs_bf16x3": alt_fp32_maxdiff,
                "note": "same job with ww_model_set_precision(WW_PRECISION_FP32); per-rank time, not max-reduced"}
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(eng, pcm0, float(os.environ.get("WW_BENCH_CPU_SECONDS", "10")))
        elif not args.no_cpu_baseline:
            line["cpu_baseline"] = None
        print(json.dumps(line))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
