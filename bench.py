#!/usr/bin/env python3
"""Headline benchmark: audio frames/s of the wake-word hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--model crnn|wavenet] [--clips 256]

One step = one pass of the whole hot path over one batch of synthetic 16 kHz clips that is
already resident in HBM: int16 PCM [clips, 24000] -> log-mel -> one zero-padded window per
clip -> encode + detect -> posteriors [clips, n_out]   (BASELINE.json configs[1] for CRNN,
configs[2] for Wavenet; 1 audio frame = one 10 ms hop = 160 samples, 150 per 1.5 s clip).

N > 1.  Either something else started the ranks (torch.distributed.run sets WORLD_SIZE), or this
process does it itself: with --gpus N and no WORLD_SIZE in the environment the parent - which
never imports torch and never touches a GPU - starts N child ranks of this same file
(subprocess, RANK/LOCAL_RANK/WORLD_SIZE/MASTER_ADDR/MASTER_PORT in their environment), relays
rank 0's single JSON line and exits non-zero if any child failed.  Every rank owns its own
shard of clips (weak scaling, no data-path collective); the only exchange is the posterior
gather (all_gather of the K*clips*n_out floats each rank produced, RCCL over xGMI), done once
at the end of every timed region, as the offline evaluator does after its shard is finished.

Timing.  W untimed warm-up steps (at least one per pipelined context), then the region of
EXACTLY K steps - barrier + synchronize on both sides, MAX over ranks - is repeated --repeats
times; `value` / `ms_per_step` are those of the MEDIAN region and `timed_regions` holds
min / median / max, so that a short K is not a single sample.  The input rotates over
--rotate (24) distinct resident batches whatever K is: 24 x 12.3 MB = 295 MB > the 256 MiB
Infinity Cache, so a step's PCM comes from HBM.

Rank 0 prints ONE JSON line (schema in the task contract) that also carries
  roofline      - dominant kernel, HIP events around every launch (ww_profile_enable) in a
                  separate pass of the same steps, priced per DESIGN.md section 4
  cpu_baseline  - the C restatement in oracle/ (NOT TFLite) on this box's host cores, bounded
  single_stream - the same job on ONE context (strict batch-256 latency chain, --pipeline 1)
  wavenet       - BASELINE configs[2]: split-bf16 MFMA and the fp32-MFMA parity mode
  streaming     - BASELINE configs[4]: 128 streams per GPU in lock step, tick latency p50/p99
  eval_testset  - BASELINE configs[0]/[3]: the reference evaluator's own flow on the 2,048-clip
                  stand-in of the hey-snips test split - positives utterance-sharded, the ONE joined
                  negative stream cut into contiguous posterior ranges over the N ranks - FRR @ 0.5
                  FA/h (the second half of the metric) next to the C-oracle value of the same flow
                  (N = 1 only).
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

SAMPLES = 24000          # 1.5 s @ 16 kHz
FRAMES_PER_CLIP = 150    # audio frames (10 ms hops) per clip
PEAK_F32_MFMA = 157.3e12  # MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32 / 32x32x2, 256 CUs
PEAK_HBM = 8.0e12
PEAK_BF16_MFMA = 2.5e15   # dense bf16, MI355X_MICROARCH.md
PEAK_F64_VALU = 78.6e12   # fp64 vector FMA peak
N_SIMD = 1024             # 256 CUs x 4
VALU_CYCLES_PER_INST = 3.7  # measured: tools/pmc_calib.hip (v_fma_f32, >= 2 waves per SIMD); fp64 ops issue at the same rate (tools/valu_probe.hip)


# ------------------------------------------------------------------------------------------------
# parent: start the ranks (no torch, no GPU call in this process)
# ------------------------------------------------------------------------------------------------
def launch_ranks(n: int) -> int:
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    out0, _ = procs[0].communicate()
    codes = [procs[0].returncode] + [p.wait() for p in procs[1:]]
    lines = [ln for ln in (out0 or "").splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    if any(codes) or not lines:
        sys.stderr.write(f"bench.py: rank exit codes {codes}\n")
        return next((c for c in codes if c), 1)
    return 0


# ------------------------------------------------------------------------------------------------
def synth_pcm(rng, n_clips):
    """Gaussian noise sigma=2000 LSB + linear chirp 200->4000 Hz at 8000 LSB."""
    import numpy as np
    t = np.arange(SAMPLES) / 16000.0
    phase = 2 * np.pi * (200.0 * t + 0.5 * (4000.0 - 200.0) / 1.5 * t * t)
    x = rng.normal(0.0, 2000.0, (n_clips, SAMPLES)) + 8000.0 * np.sin(phase)[None, :]
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def kernel_source_sha():
    """sha256 over the kernel sources the library is built from: PMC figures measured on other sources are not evidence for
    this build (there is no .git on the GPU box, so the sources themselves are the key)."""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(PKG, "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".h")):
            text = open(os.path.join(d, f), "rb").read()
            if (f.endswith(".hip") and b"__global__" not in text) or (f.endswith(".h") and b"__device__" not in text and b"__global__" not in text):
                continue  # host code only (uploader.hip, host_stage.h): no kernel changes with it
            h.update(f.encode())
            h.update(text)
    return h.hexdigest()[:16]


PMC_FILE = os.path.join("profiles", "r06", "pmc_counters.json")


def pmc_counters():
    """profiles/r06/pmc_counters.json (tools/pmc_collect.py: rocprofv3 --pmc passes, mean per launch and kernel) if it
    was measured on THESE kernel sources, else None."""
    try:
        pmc = json.load(open(os.path.join(ROOT, PMC_FILE)))
    except (OSError, ValueError):
        return None
    return pmc if pmc.get("source_sha") == kernel_source_sha() else None


def kernel_work(eng, n_clips):
    """name -> (bound, algorithmic HBM bytes, algorithmic FLOPs (2*MAC)) per launch; DESIGN.md section 4."""
    nf = (SAMPLES - 512) // 160 + 1
    out = {}
    fe_bytes = n_clips * (SAMPLES * 2 + nf * 40 * 4)
    fe_flops = n_clips * nf * 13.9e3
    out["logmel_rows_kernel"] = ("hbm", fe_bytes, fe_flops)  # the fp64 front end (round 4's kernel, under its own name since round 5)
    out["logmel_kernel<f64>"] = ("hbm", fe_bytes, fe_flops)  # (-DWW_FE_OLD=1 builds: rounds 1-3's fp64 kernel)
    out["logmel_kernel<f32>"] = ("hbm", fe_bytes, fe_flops)
    if eng.is_crnn:
        c = eng.bundle.crnn
        M = c.out_t * c.out_f
        K = c.conv_w.shape[1] * c.conv_w.shape[2]
        H = c.units
        # the whole model in one kernel: mel window in, posterior out; the 622 KB of weights are read once per XCD
        flops = 2.0 * (M * K * 32 + c.out_t * (c.out_f * 32) * 6 * H + c.out_t * 2 * H * 6 * H + 2 * c.out_t * 2 * 3 * H * H
                       + 64 * 64 + 64 * c.n_out)
        out["crnn_fused_kernel"] = ("mfma", n_clips * (eng.window * 40 * 4 + c.n_out * 4) + 8 * 622724, n_clips * flops)
    else:
        w = eng.bundle.wavenet
        macs = w.n_frames * (w.n_mel * w.channels) + sum(
            w.n_frames * (3 * w.channels * 2 * w.channels + w.channels * ((w.channels if b.w_res is not None else 0) + w.skip_channels))
            for b in w.blocks) + w.n_frames * (w.skip_channels * w.skip_channels + w.skip_channels * eng.n_out)
        out["wavenet_kernel"] = ("mfma", n_clips * (eng.window * 40 * 4 + eng.n_out * 4), n_clips * 2.0 * macs)
        out["wavenet_kernel<bf16x3>"] = out["wavenet_kernel"]
    return out


def granted_cpus():
    """CPUs this job may actually use: the affinity mask, capped by the cgroup CPU quota when there is one."""
    try:
        n = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        n = os.cpu_count() or 1
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            elif int(txt[0]) > 0:
                period = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                n = min(n, max(1, int(int(txt[0]) / period + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(eng, pcm_sample, budget_s=10.0):
    """Time the C restatement (oracle/ww_oracle.c) on a bounded sample, as SURVEY 8(d) defines the CPU baseline when TFLite
    is absent: ONE thread (the reference path is one TFLite interpreter, default threads = 1, batch 1:
    spokestack/models/tensorflow.py:26-28) and all the cores this job is granted."""
    import numpy as np
    from oracle import cpu as ocpu
    ora = ocpu.CpuOracle(eng.blob)
    nproc, granted = os.cpu_count() or 1, granted_cpus()
    # a 1-GPU box grants this job 16 CPUs of many more logical ones: use exactly that share
    all_threads = max(1, min(16, granted))

    def one_pass(clips):
        wins = np.zeros((len(clips), eng.window, 40), np.float32)
        for i, c in enumerate(clips):
            mel = ora.logmel(c)
            n = min(len(mel), eng.window)
            wins[i, :n] = mel[:n]
        return ora.forward(wins)

    def timed(threads, sample, budget):
        used = ocpu.set_threads(threads)
        one_pass(sample[:used])  # warm up (thread pool, tables)
        t0 = time.perf_counter()
        done = 0
        while True:
            one_pass(sample)
            done += len(sample)
            el = time.perf_counter() - t0
            if el > budget:
                return used, done, el

    t_all, n_all, el_all = timed(all_threads, pcm_sample, 0.6 * budget_s)
    t_one, n_one, el_one = timed(1, pcm_sample[:16], 0.4 * budget_s)
    return {
        "value": n_all * FRAMES_PER_CLIP / el_all,
        "unit": "audio frames/s",
        "cores": t_all,
        "kind": "port",
        "sample": f"{n_all} clips x 1.5 s ({len(pcm_sample)}-clip sample of the same synthetic batch, repeated), "
                  f"{el_all:.1f} s wall on {t_all} threads; C restatement oracle/ww_oracle.c with OpenMP over clips, NOT TFLite",
        "one_thread": {"value": n_one * FRAMES_PER_CLIP / el_one, "unit": "audio frames/s", "cores": t_one,
                       "sample": f"{n_one} clips x 1.5 s (16-clip sample, repeated), {el_one:.1f} s wall; the reference's own "
                                 "shape: one interpreter, one thread, one clip at a time"},
        "host": {"nproc": nproc, "granted_cpus": granted},
    }


class Job:
    """K steps of the clip path dealt round-robin to P contexts; inputs rotate over R resident batches;
    step k writes its posteriors to row k of one [K, clips, n_out] buffer (what the gather sends)."""

    def __init__(self, torch, engs, ctxs, d_pcm, clips, K, dist, comm_dev, world):
        self.torch, self.engs, self.ctxs, self.d_pcm, self.clips, self.K = torch, engs, ctxs, d_pcm, clips, K
        self.dist, self.comm_dev, self.world = dist, comm_dev, world
        self.n_out = engs[0].n_out
        self.d_all = torch.zeros((K, clips, self.n_out), dtype=torch.float32, device="cuda")
        self.row_ptr = [self.d_all[k].data_ptr() for k in range(K)]
        self.pcm_ptr = [t.data_ptr() for t in d_pcm]
        self.gathered = None

    def step(self, k, fp, only0=False):
        e = self.engs[0] if only0 else self.engs[k % len(self.engs)]
        e.clips_forward_dev(self.pcm_ptr[k % len(self.pcm_ptr)], self.clips, SAMPLES, self.row_ptr[k % self.K], fp)

    def sync(self):
        for c in self.ctxs:
            c.synchronize()

    def barrier(self):
        if self.dist is not None:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def warm(self, n, fp):
        for k in range(n):
            self.step(k, fp)
        self.sync()
        if self.dist is not None:  # the communicator too, outside the timed regions - and the gather alone, timed
            import numpy as np
            ts = []
            for _ in range(5):
                self.barrier()
                t0 = time.perf_counter()
                self._gather()
                ts.append(time.perf_counter() - t0)
            self.collective_ms = float(np.median(ts[1:]) * 1e3)

    def _gather(self):
        """The one exchange of the job: every rank's K x clips x n_out posteriors to every rank.  Returns when the data
        has arrived HERE, which it cannot before every rank has contributed: the gather is the region's closing barrier."""
        if self.comm_dev == "cuda":
            # RCCL: ONE flat destination [world][K][clips][n_out] (all_gather_into_tensor) - the list form copies every rank's
            # piece once more on the way out, which at 20 KB per rank is most of what the call costs
            if self.gathered is None:
                self.gathered = self.torch.empty((self.world,) + tuple(self.d_all.shape), dtype=self.d_all.dtype, device=self.d_all.device)
            self.dist.all_gather_into_tensor(self.gathered, self.d_all)
            self.torch.cuda.synchronize()
        else:
            src = self.d_all.cpu()
            if self.gathered is None:
                self.gathered = [self.torch.empty_like(src) for _ in range(self.world)]
            self.dist.all_gather(self.gathered, src)

    def region(self, fp, only0=False):
        """EXACTLY K steps between barrier + synchronize; returns (seconds: MAX over ranks, this rank's own seconds).
        The clock stops when the posterior gather has completed on this rank - no rank's gather completes before every
        rank has finished its K steps, so the MAX over ranks is the job's time without a second barrier inside the region
        (round 3 had one: an extra all-reduce round trip per 1 ms region)."""
        self.barrier()
        t0 = time.perf_counter()
        for k in range(self.K):
            self.step(k, fp, only0)
        self.sync()
        if self.dist is not None:
            self._gather()  # posterior gather, once per job
        own = el = time.perf_counter() - t0
        if self.dist is not None:
            t = self.torch.tensor([el], dtype=self.torch.float64, device=self.comm_dev)
            self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
            el = float(t.item())
        return el, own

    def regions(self, n, fp, only0=False):
        import numpy as np
        both = np.array([self.region(fp, only0) for _ in range(n)])
        ts, own = both[:, 0], both[:, 1]
        stats = {"n": int(n), "min_ms": float(ts.min() * 1e3), "p10_ms": float(np.percentile(ts, 10) * 1e3),
                 "median_ms": float(np.median(ts) * 1e3), "p90_ms": float(np.percentile(ts, 90) * 1e3), "max_ms": float(ts.max() * 1e3),
                 "timed_seconds": float(ts.sum())}
        if self.dist is not None:
            # every rank's own median region (its clock stops at ITS gather's completion): the spread says who waits for whom
            t = self.torch.tensor([float(np.median(own))], dtype=self.torch.float64, device=self.comm_dev)
            every = [self.torch.zeros_like(t) for _ in range(self.world)]
            self.dist.all_gather(every, t)
            per_rank = [float(x.item()) * 1e3 for x in every]
            stats["per_rank_median_ms"] = {"min": min(per_rank), "max": max(per_rank)}
            stats["collective_ms"] = getattr(self, "collective_ms", None)
        return stats, float(np.median(ts))


def profile_pass(job, ctx, K, fp, min_launches=200, chunk_max=25):
    """HIP events around every launch (ww_profile_enable) on ONE context, in chunks of <= 25 steps and at least 200 steps in
    all; per kernel the MEDIAN over the chunks of the chunk's mean duration.  (A plain mean over the driver's 20 steps is owned
    by one stalled launch - a clock ramp, a page migration: seen once as a 9.7 ms launch among nineteen of 33 us.)"""
    import numpy as np
    chunk = max(1, min(K, chunk_max))
    n_chunks = max(3, -(-max(K, min_launches) // chunk))
    means = {}
    ctx.profile(True)
    k = 0
    for _ in range(n_chunks):
        for _ in range(chunk):
            job.step(k, fp, only0=True)
            k += 1
        for name, rec in ctx.profile_read().items():
            means.setdefault(name, []).append(rec["total_ms"] / max(rec["calls"], 1))
    ctx.profile(False)
    return {name: {"calls": n_chunks * chunk, "total_ms": float(np.median(v)) * n_chunks * chunk, "chunks": len(v)} for name, v in means.items()}


def throughput(world, K, clips, seconds):
    return world * K * clips * FRAMES_PER_CLIP / seconds


def roofline_of(eng, prof, clips, fast_frontend, step_s=None):
    kw = kernel_work(eng, clips)
    per_kernel = {}
    dom, dom_ms = None, -1.0
    for name, rec in prof.items():
        avg = rec["total_ms"] / max(rec["calls"], 1)
        per_kernel[name] = round(avg * 1e3, 3)  # microseconds
        if name in kw and avg > dom_ms:
            dom, dom_ms = name, avg
    bound, nbytes, flops = kw[dom]
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
    if bound == "hbm" and not fast_frontend:
        # the front end is priced against HBM as SURVEY 8(d) defines it; what actually bounds it is the fp64
        # vector ALU (Hann product, two radix-16 DFTs and the untangling are all float64 like the reference)
        tf64 = flops / (dom_ms * 1e-3) / 1e12
        roof["fp64_valu"] = {"achieved_TFLOPs": tf64, "peak_TFLOPs": PEAK_F64_VALU / 1e12, "frac": tf64 * 1e12 / PEAK_F64_VALU,
                             "note": "13.9 kFLOP per frame (SURVEY 8d); mostly adds, so half of the FMA peak is the ceiling"}
    roof["traffic"] = None
    pmc = pmc_counters()
    rec = (pmc or {}).get("workloads", {}).get(f"clips{clips}", {}).get(dom)
    if rec and "FETCH_SIZE" in rec and "WRITE_SIZE" in rec:
        roof["traffic"] = (2 * rec["FETCH_SIZE"] + rec["WRITE_SIZE"]) * 1024
        roof["traffic_note"] = ("2 x FETCH_SIZE + WRITE_SIZE (KiB; gfx950 reports half of 16-byte-per-lane streaming reads: "
                                "MI355X_MICROARCH.md), rocprofv3 --pmc in separate passes, mean bytes per launch "
                                f"({PMC_FILE}, kernel sources {pmc['source_sha']})")
        roof["algorithmic_bytes"] = nbytes
        if dom.startswith("crnn_fused_kernel"):
            # SURVEY 8(d) counts the 622,724 B of weights ONCE; each of the 8 XCDs has an L2 of its own and fetches them once,
            # which is what HBM-side traffic can reach at best - both figures, so that the ratio to `traffic` is unambiguous
            roof["algorithmic_bytes_weights_once"] = nbytes - 7 * 622724
            roof["algorithmic_bytes_note"] = ("algorithmic_bytes = windows in + posteriors out + the weights once PER XCD L2 (8 x 622,724 B); "
                                              "algorithmic_bytes_weights_once = SURVEY 8(d)'s definition (weights once per launch)")
    elif pmc is None:
        roof["traffic_note"] = f"no PMC pass on these kernel sources ({PMC_FILE} absent or of another source_sha)"
    # how busy the shared vector / matrix datapath is over a STEP (the fp32 MFMA and the vector ALU of a SIMD do not
    # co-execute on gfx950: tools/pmc_calib.hip, profiles/r04/README.md): per kernel, busy cycles per SIMD =
    # SQ_VALU_MFMA_BUSY_CYCLES / 1024 + 3.7 x (SQ_INSTS_VALU - SQ_INSTS_MFMA) / 1024 over the kernel's own SQ_BUSY_CYCLES / 32
    # (all from one PMC pass, so the clock drops out), weighted with the kernels' live durations over the live step time
    roof["step_datapath_busy"] = None
    wl = (pmc or {}).get("workloads", {}).get(f"clips{clips}", {})
    if step_s and all(k in wl and "SQ_BUSY_CYCLES" in wl[k] and "SQ_INSTS_VALU" in wl[k] for k in prof if k in kw):
        parts, busy_s = {}, 0.0
        for name, rec in prof.items():
            if name not in kw:
                continue
            c = wl[name]
            cyc = (c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) + VALU_CYCLES_PER_INST * (c["SQ_INSTS_VALU"] - c.get("SQ_INSTS_MFMA", 0.0))) / N_SIMD
            b = cyc / (c["SQ_BUSY_CYCLES"] / 32.0)
            t = rec["total_ms"] / max(rec["calls"], 1) * 1e-3
            parts[name] = {"busy_frac_in_kernel": b, "matrix_frac_in_kernel": c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / N_SIMD / (c["SQ_BUSY_CYCLES"] / 32.0),
                           "kernel_us": t * 1e6}
            busy_s += b * t
        roof["step_datapath_busy"] = {"value": busy_s / step_s, "step_us": step_s * 1e6, "kernels": parts,
                                      "method": "sum over the step's kernels of (matrix-busy + vector-issue cycles per SIMD, PMC) / (kernel cycles, "
                                                "PMC) x live kernel duration, over the live pipelined step time; 3.7 cycles per vector "
                                                "instruction and 32 per v_mfma_f32_16x16x4_f32 as calibrated by tools/pmc_calib.hip"}
    roof["kernel"] = dom
    roof["kernel_avg_us"] = round(dom_ms * 1e3, 3)
    roof["all_kernels_avg_us"] = per_kernel
    # every priced kernel of the step against its own bound (the dominant one is the object itself)
    every = {}
    for name, rec in prof.items():
        if name not in kw:
            continue
        b, nb, fl = kw[name]
        t = rec["total_ms"] / max(rec["calls"], 1) * 1e-3
        if b == "hbm":
            every[name] = {"bound": "hbm", "achieved": nb / t / 1e9, "peak": PEAK_HBM / 1e9, "unit": "GB/s", "frac": nb / t / PEAK_HBM}
            if not fast_frontend and fl:
                # SURVEY 8(d) prices the front end against HBM; what its instructions run on is the fp64 vector ALU
                every[name]["fp64_valu"] = {"achieved_TFLOPs": fl / t / 1e12, "peak_TFLOPs": PEAK_F64_VALU / 1e12,
                                            "frac": fl / t / PEAK_F64_VALU,
                                            "note": "13.9 kFLOP per frame (SURVEY 8d), mostly adds: half of the FMA peak is the ceiling; "
                                                    "vector ALU 55 % busy over the kernel (pmc_counters.json: 3.7 cycles x non-MFMA vector instructions per SIMD over SQ_BUSY_CYCLES / 32)"}
        elif name.endswith("<bf16x3>"):
            every[name] = {"bound": "mfma", "achieved": 3.0 * fl / t / 1e12, "peak": PEAK_BF16_MFMA / 1e12, "unit": "TFLOP/s",
                           "frac": 3.0 * fl / t / PEAK_BF16_MFMA}
        else:
            every[name] = {"bound": "mfma", "achieved": fl / t / 1e12, "peak": PEAK_F32_MFMA / 1e12, "unit": "TFLOP/s",
                           "frac": fl / t / PEAK_F32_MFMA}
    roof["kernels"] = every
    roof["method"] = ("HIP events around each launch (ww_profile_enable) on the launching stream, right after the timed regions: >= 200 "
                      "single-stream steps in chunks of <= 25, per kernel the median over the chunks of the chunk's mean duration")
    return roof


def stream_leg(torch, np, dist, comm_dev, local_rank, S, ticks, warm=100):
    """BASELINE configs[4] per GPU: S streams x int16[320] per 20 ms tick, is_speech = 1 (2 posteriors per
    stream and tick); latency = tick submitted on the host -> posteriors visible on the host."""
    from wwhip.activation_timeout import ActivationTimeoutBank
    from wwhip.engine import Engine, StreamBank, frontend_params
    from wwhip.vad import VadBank
    from wwhip.wakeword import WakewordBank
    out = {"streams_per_gpu": S, "ticks": ticks, "warmup_ticks": warm,
           "pipeline_note": "pipeline = the same tick at the plugin surface (spokestack/pipeline.py:25-28 with demo.py's stage list, "
                            "for S streams): SpeechPipelineBank.step() over VadBank -> WakewordBank -> ActivationTimeoutBank on a ContextBank: the input "
                            "source's read(), the batch classifier, ONE library call for the three stages' passes (ww_pipeline_bank_step = "
                            "ww_vad_bank_step + ww_stream_step_trigger + ww_timeout_bank_step in stage order), raw VAD decision = "
                            "speech on every stream (the worst case: every stream owes 2 posteriors), threshold 0.5; "
                            "over_tick_us = pipeline p50 - StreamBank.step p50 of the same bank kind, measured back to back",
           "note": "per-tick latency, host frames in -> host posteriors out; MAX over ranks of each rank's percentile; a tick is ONE "
                   "kernel launch (round 5: the front end of a stream's new frames runs inside the workgroups of its new windows - "
                   "crnn_stream_kernel<FE>: 3 of the 19 time positions of a window are new per mel row, the other 16 projected rows "
                   "come from a per-stream ring; wavenet_kernel<tick>) and the host polls {posterior, tick} pairs the heads store into "
                   "page-locked memory instead of hipStreamSynchronize; host_phases_us = ww_stream_timeline (rank 0's mean per tick: "
                   "plan, frames into the pinned block, launch 1, launch 2, wait, copy-out) + python_wrapper_us = step mean - their sum"}
    rng = np.random.default_rng(5)
    frames = np.clip(rng.normal(0, 2500, (64, S, 320)), -32768, 32767).astype(np.int16)
    speech = np.ones(S, np.uint8)
    for name, prec in (("CRNN", "fp32"), ("Wavenet", "bf16x3"), ("Wavenet", "fp32")):
        eng = Engine(os.path.join(PKG, "assets", "tf_lite_models", name), device=local_rank, precision=prec)
        bank = StreamBank(eng, S)
        for t in range(warm):
            bank.step(frames[t % 64], speech)
        bank.timeline(reset=True)
        lat = np.empty(ticks)
        n_post = 0
        for t in range(ticks):
            t0 = time.perf_counter()
            _, n = bank.step(frames[t % 64], speech)
            lat[t] = time.perf_counter() - t0
            n_post += int(n.sum())
        tl = bank.timeline()
        tl.pop("ticks", None)
        stats = [float(np.percentile(lat, 50) * 1e3), float(np.percentile(lat, 99) * 1e3), float(lat.mean() * 1e3),
                 float(np.percentile(lat, 99.9) * 1e3)]
        own_mean_us = stats[2] * 1e3
        if dist is not None:
            t = torch.tensor(stats, dtype=torch.float64, device=comm_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            stats = [float(v) for v in t.tolist()]
        out[name.lower() if prec == "fp32" else f"{name.lower()}_{prec}"] = {
            "p50_ms": stats[0], "p99_ms": stats[1], "p99.9_ms": stats[3], "mean_ms": stats[2], "posteriors_per_tick": n_post / ticks,
            "realtime_factor": 0.020 / (stats[2] * 1e-3), "host_phases_us": {k: round(v, 3) for k, v in tl.items()},
            "python_wrapper_us": round(own_mean_us - sum(tl.values()), 3)}
        bank.close()
        # ---- the same tick at the plugin surface: three banked stages on a ContextBank
        from wwhip.pipeline import SpeechPipelineBank
        raw = np.ones(S, bool)

        class Source:  # pipeline.step() reads its frames from the input source (spokestack/pipeline.py:25)
            frame = frames[0]

            def read(self):
                return self.frame

            def start(self):
                pass

            stop = close = start

        src = Source()
        wake = WakewordBank(S, posterior_threshold=0.5, bank=StreamBank(eng, S, frontend_params(32767.0, True, 0.0, 160, True)))
        pipe = SpeechPipelineBank(src, [VadBank(S, classifier=lambda f: raw), wake, ActivationTimeoutBank(S)], S)
        events = []
        pipe.event(lambda c: events.append(1), name="activate")
        pipe.event(lambda c: events.append(0), name="deactivate")
        pipe.start()
        for t in range(warm):
            src.frame = frames[t % 64]
            pipe.step()
        plat = np.empty(ticks)
        for t in range(ticks):
            src.frame = frames[t % 64]
            t0 = time.perf_counter()
            pipe.step()
            plat[t] = time.perf_counter() - t0
        pstats = [float(np.percentile(plat, 50) * 1e3), float(np.percentile(plat, 99) * 1e3), float(plat.mean() * 1e3)]
        if dist is not None:
            t = torch.tensor(pstats, dtype=torch.float64, device=comm_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            pstats = [float(v) for v in t.tolist()]
        key = name.lower() if prec == "fp32" else f"{name.lower()}_{prec}"
        out[key]["pipeline"] = {"p50_ms": pstats[0], "p99_ms": pstats[1], "mean_ms": pstats[2],
                                "over_tick_us": round((pstats[0] - out[key]["p50_ms"]) * 1e3, 3),
                                "activations": int(sum(events)), "deactivations": len(events) - int(sum(events))}
        pipe.stop()
        wake.close()
        eng.close()
    return out


def eval_at_scale(torch, np, dist, comm_dev, rank, world, eng, n_wake, with_oracle):
    """BASELINE configs[3] at the SIZE it names: the reference evaluator's flow (utils/evaluate_models.py:281-326) over a
    stand-in as large as the hey-snips test split - 2,529 wake-word clips through the never-reset Filter + the first 2,529
    other clips joined into ONE ~1.2 h stream (concatenate_FA joins exactly num_wakewords of them, :299) - sharded over the
    ranks, with where the time goes: host phases (plan, slicing = staging the samples, upload, gather, sweep; staging runs beside the kernels) against the
    kernels' own time (HIP events, ww_profile_read).  Next to it the C oracle on a bounded sample of the same flow."""
    from wwhip.evaluate import synth_testset_scaled, evaluate_reference_flow_sharded, SHARE_ONLY
    clips, labels = synth_testset_scaled(n_wake, n_wake)
    SERIAL = ("prepare", "plan", "gather", "sweep", "d2h", "profile_read")  # what every rank repeats whatever the world size

    def passes(cl, lb, n, rk, wd, comm):
        """n timed passes WITHOUT kernel events (host phases only: perf_counter reads), sorted by time; then one more pass with HIP
        events around every launch, whose kernel table (device_ms, kernels_ms) is attached to every timed pass - round 5 timed the
        profiled passes themselves (two events per launch and their read-back inside the clock)."""
        runs = []
        for attempt in range(n + 1):  # the first pass sizes staging buffers and workspaces
            if dist is not None and comm != SHARE_ONLY:
                dist.barrier()
            torch.cuda.synchronize()
            tm = {"kernel_times": False}
            t0 = time.perf_counter()
            r = evaluate_reference_flow_sharded(eng, cl, lb, rk, wd, comm, timing=tm)
            el = time.perf_counter() - t0
            if attempt:
                runs.append((el, tm, r))
        if dist is not None and comm != SHARE_ONLY:
            dist.barrier()
        torch.cuda.synchronize()
        tk = {}
        t0 = time.perf_counter()
        evaluate_reference_flow_sharded(eng, cl, lb, rk, wd, comm, timing=tk)
        el_k = time.perf_counter() - t0
        for _, tm, _ in runs:
            tm["device_ms"], tm["kernels_ms"], tm["profiled_pass_seconds"] = tk["device_ms"], tk.get("kernels_ms"), el_k
        runs.sort(key=lambda x: x[0])
        return runs

    def describe(runs, cl, n_w):
        el, tm, r = runs[0]
        audio_s = (sum(len(c) for c in cl[:n_w]) + r["hours"] * 3600 * 16000) / 16000.0
        ph_ms = {k: v * 1e3 for k, v in tm.items() if isinstance(v, float) and k not in ("device_ms", "profiled_pass_seconds")}
        # the serial part of every pass; the MEDIAN is reported (a 15 ms job on a shared host: one pass in three or four catches
        # a scheduling hiccup of a few hundred microseconds in one of its Python phases)
        serials = sorted(sum(v * 1e3 for k, v in t.items() if k in SERIAL and isinstance(v, float)) for _, t, _ in runs)
        serial = serials[len(serials) // 2]
        return r, audio_s, {
            "seconds_host_pcm_in_to_curves_out": el, "median_seconds": runs[len(runs) // 2][0], "passes": len(runs),
            "audio_hours": audio_s / 3600.0, "audio_frames_per_s": audio_s * 100.0 / el, "realtime_factor": audio_s / el,
            "windows": r["windows"], "device_ms": tm["device_ms"], "kernels_ms": tm.get("kernels_ms"), "host_phases_ms": ph_ms,
            "chunks": tm.get("chunks"), "host_share": 1.0 - tm["device_ms"] * 1e-3 / el, "profiled_pass_seconds": tm["profiled_pass_seconds"],
            "serial_ms": serial, "serial_ms_min": serials[0],
            "predicted_seconds": {str(w): serial * 1e-3 + (el - serial * 1e-3) / w for w in (2, 4, 8)},
            "predicted_efficiency_8_ranks": el / (8 * (serial * 1e-3 + (el - serial * 1e-3) / 8))}

    runs = passes(clips, labels, 9, rank, world, comm_dev)
    one_of_8 = x16 = x16_of_8 = None
    if world == 1:
        # what ONE rank of eight costs, measured: rank 0's share of a world of 8 on this GPU, no communicator (its peers' slots
        # stay zero; the posterior gather - one RCCL all_gather per leg, latency - is what this leaves out)
        one_of_8 = passes(clips, labels, 9, 0, 8, SHARE_ONLY)
        # the same flow over 16 x the clips (each clip object referenced 16 times: ~38 h of audio, 5.7 GB uploaded): a device
        # time large enough to show a curve
        clips16 = clips[:n_wake] * 16 + clips[n_wake:] * 16
        labels16 = np.concatenate((np.ones(16 * n_wake, np.uint8), np.zeros(16 * (len(clips) - n_wake), np.uint8)))
        x16 = passes(clips16, labels16, 3, 0, 1, None)
        x16_of_8 = passes(clips16, labels16, 3, 0, 8, SHARE_ONLY)
    if rank != 0:
        return None
    r, audio_s, out = describe(runs, clips, n_wake)
    el = out["seconds_host_pcm_in_to_curves_out"]
    out = {"workload": f"{n_wake} wake-word clips (file by file, C2 carry) + the first {n_wake} other clips joined by 100 ms of silence "
                       f"into one {r['hours']:.2f} h stream; synthetic clips 0.8-2.5 s (float32 generator, seed 4321), CRNN_softmax",
           "world_size": world, **out,
           "note": "rank 0's clock and phases of the FASTEST of nine passes (median beside it; three at 16 x; serial_ms = the median over the passes), timed WITHOUT kernel events; device_ms / "
                   "kernels_ms come from one more pass with HIP events around every launch (profiled_pass_seconds); every rank stages, uploads and runs "
                   "only its share; host_share = 1 - device_ms / "
                   "seconds; the share goes to the GPU in chunks of up to ~26 min of audio: this thread plans a chunk (prepare / plan / slicing: "
                   "which samples of which clips), submits it to the library's uploader (ww_uploader: its threads write the chunk "
                   "into page-locked memory once and start the upload) and enqueues the kernels over the chunk before, behind its "
                   "upload (upload_wait = waiting for the uploader, device_wall = launches + the final wait for the GPU).  "
                   "serial_ms = the phases every rank repeats whatever the world size (prepare + plan: lengths of ALL clips, every "
                   "rank's share; gather; sweep; d2h; profile_read); predicted_seconds[w] = serial + (seconds - serial) / w from "
                   "THIS pass's phases (no collective latency in it)",
           "frr_at_0.5_fa_per_hour": r["frr_at_0.5_fa_per_hour"], "fa_count_at_threshold_0.5": int(r["fa_count"][0]),
           "posterior_checksum": r["posterior_checksum"]}
    if one_of_8 is not None:
        _, _, d8 = describe(one_of_8, clips, n_wake)
        out["one_rank_of_8_measured"] = {
            "seconds": d8["seconds_host_pcm_in_to_curves_out"], "median_seconds": d8["median_seconds"], "device_ms": d8["device_ms"],
            "host_phases_ms": d8["host_phases_ms"], "chunks": d8["chunks"],
            "speedup_vs_one_rank": el / d8["seconds_host_pcm_in_to_curves_out"],
            "efficiency_8_ranks": el / (8 * d8["seconds_host_pcm_in_to_curves_out"]),
            "note": "rank 0's share of a world of 8 run alone on this GPU (no communicator: the two posterior all_gathers are "
                    "left out, their latency is timed_regions.collective_ms of an N > 1 line); the longest-first deal and the "
                    "equal posterior ranges make every rank's share the same size within a clip"}
        r16, _, d16 = describe(x16, clips16, 16 * n_wake)
        _, _, d16_8 = describe(x16_of_8, clips16, 16 * n_wake)
        out["at_scale_x16"] = {
            "workload": f"the same flow over 16 x the clips ({16 * n_wake} + {16 * n_wake}; every clip object referenced 16 times), "
                        f"{r16['hours']:.1f} h joined stream", **d16,
            "one_rank_of_8_measured": {"seconds": d16_8["seconds_host_pcm_in_to_curves_out"], "device_ms": d16_8["device_ms"],
                                       "host_phases_ms": d16_8["host_phases_ms"],
                                       "speedup_vs_one_rank": d16["seconds_host_pcm_in_to_curves_out"] / d16_8["seconds_host_pcm_in_to_curves_out"],
                                       "efficiency_8_ranks": d16["seconds_host_pcm_in_to_curves_out"] / (8 * d16_8["seconds_host_pcm_in_to_curves_out"])}}
    if with_oracle:
        # the C oracle (the checker) on a bounded sample of the same flow: the first 32 wake-word clips and the first 120 s of
        # the joined stream, on the granted cores; its time scaled by the audio ratio is what the whole flow would take
        from oracle import cpu as ocpu
        from wwhip.evaluate import StreamPlan, join_negatives
        ora = ocpu.CpuOracle(eng.blob)
        ocpu.set_threads(max(1, min(16, granted_cpus())))
        pidx = eng.posterior_index
        wake = clips[:32]
        stream = join_negatives(clips[n_wake:], n_wake)[:120 * 16000]
        t0 = time.perf_counter()
        plan = StreamPlan([len(stream)], eng.window)
        padded = np.zeros(int(plan.padded[0]), np.int16)
        padded[8000:8000 + len(stream)] = stream
        o_neg = ora.slide_forward(ora.logmel(padded, 32768.0, False), 2)[:plan.total, pidx]
        planw = StreamPlan([len(c) for c in wake], eng.window)
        whole = np.zeros(int(planw.padded.sum()), np.int16)
        for k, c in enumerate(wake):
            whole[planw.pos[k] + 8000: planw.pos[k] + 8000 + len(c)] = c
        mel = ora.logmel(whole, 32768.0, False)
        o_pos = np.array([ora.slide_forward(mel[planw.F[k]: planw.F[k] + planw.n_frames[k]], 2)[:planw.n_win[k], pidx].max()
                          for k in range(len(wake))], np.float32)
        o_el = time.perf_counter() - t0
        s_audio = (len(stream) + sum(len(c) for c in wake)) / 16000.0
        out["oracle_sample"] = {"kind": "oracle/ww_oracle.c (C restatement, NOT TFLite), OpenMP on the granted cores", "seconds": o_el,
                                "audio_seconds": s_audio, "extrapolated_seconds_for_the_whole_flow": o_el * audio_s / s_audio,
                                # (the sample's last windows look at its own end padding instead of the stream going on)
                                "max_abs_posterior_diff": float(max(np.abs(o_neg[:-100] - r["negatives"][:len(o_neg) - 100]).max(),
                                                                    np.abs(o_pos - r["positives"][:len(o_pos)]).max()))}
    return out


def eval_leg(torch, np, dist, comm_dev, rank, world, local_rank, n_clips, with_oracle, n_scale=2529):
    """BASELINE configs[0] stand-in / configs[3]: the reference evaluator's own flow (utils/evaluate_models.py main()) on
    2,048 synthetic labelled clips, sharded over the ranks: wake-word clips file by file through one never-reset Filter
    (utterance-sharded), the first num_wakewords other clips joined by 100 ms of silence into ONE negative stream that is
    cut into `world` contiguous posterior ranges; posterior gather; rank 0 smooths + sweeps: FRR @ 0.5 FA/h.
    Next to it the per-clip variant of round 2 (every clip on its own grid, all negatives)."""
    from wwhip.evaluate import (synth_testset, evaluate_testset_sharded, evaluate_reference_flow_sharded, frr_at_fa,
                                join_negatives, StreamPlan)
    from wwhip.models import engine_for
    eng = engine_for(os.path.join(PKG, "assets", "tf_lite_models", "CRNN_softmax"), local_rank)
    clips, labels = synth_testset(n_clips)

    def timed(fn):
        els = []
        for attempt in range(6):  # the first pass sizes page-locked slots and workspaces; the fastest AND the median of the next five
            if dist is not None:  # (a 3 ms job on a shared host: single passes were seen between 2.4 and 10 ms box to box)
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            r = fn()
            if attempt:
                els.append(time.perf_counter() - t0)
        els.sort()
        return r, els[0], els[len(els) // 2]

    r, el, el_med = timed(lambda: evaluate_reference_flow_sharded(eng, clips, labels, rank, world, comm_dev))
    pc, el_pc, el_pc_med = timed(lambda: evaluate_testset_sharded(eng, clips, labels, rank, world, comm_dev))
    fast = evaluate_reference_flow_sharded(eng, clips, labels, rank, world, comm_dev, precise=False)  # fp32-FFT front end
    at_scale = eval_at_scale(torch, np, dist, comm_dev, rank, world, eng, n_scale, with_oracle) if n_scale > 0 else None
    if rank != 0:
        return None
    lab = labels.astype(bool)
    wake = [c for c, l in zip(clips, lab) if l]
    other = [c for c, l in zip(clips, lab) if not l]
    stream = join_negatives(other, len(wake))
    audio_frames = (len(stream) + sum(len(c) for c in wake)) // 160
    res = {"workload": f"{len(clips)} synthetic clips 0.8-2.5 s (SURVEY 8d cfg 1 stand-in, seed 1234); model CRNN_softmax "
                       "(wwdetect/CRNN/models/Arik_CRNN_data_original) IN PLACE OF configs[0]'s tf_lite_models/CRNN: the evaluator "
                       "indexes [0][0][1] (utils/evaluate_models.py:80) and the shipped CRNN's sigmoid head has width 1 - quirk C1, the "
                       "script cannot run on it; the reference "
                       "evaluator's flow: wake-word clips through one never-reset Filter (0.5 s zero padding, C2 carry), the first "
                       "num_wakewords other clips joined by 100 ms of silence into ONE stream slid over continuously (hop 2, one "
                       "inference per 20 ms chunk), 30-tap smoothing, 100 thresholds",
           "flow": "utils/evaluate_models.py main(): concatenate_FA + get_posterior x 2 + plot_FRR_FAR",
           "sharding": "positives: whole files longest-first round-robin; negative stream: contiguous posterior ranges "
                       "(each rank re-reads a T-2-frame overlap); posterior all_gather; rank 0 sweeps",
           "world_size": world, "num_wakewords": r["num_wakewords"], "negative_clips_joined": r["negative_clips_joined"],
           "seconds_host_pcm_in_to_curves_out": el, "median_seconds": el_med,
           "timing": "fastest (seconds) and median (median_seconds) of five passes after one warm-up pass", "audio_frames_per_s": audio_frames / el,
           "windows": r["windows"], "negative_hours": r["hours"],
           "frr_at_0.5_fa_per_hour": r["frr_at_0.5_fa_per_hour"], "fa_count_at_threshold_0.5": int(r["fa_count"][0]),
           "posterior_checksum": r["posterior_checksum"],
           "fast_profile_fp32_fft": {"fa_counts_identical": bool(np.array_equal(fast["fa_count"], r["fa_count"])),
                                     "frr_identical": bool(np.array_equal(fast["frr"], r["frr"])),
                                     "max_abs_posterior_diff": float(max(np.abs(fast["negatives"] - r["negatives"]).max(),
                                                                         np.abs(fast["positives"] - r["positives"]).max())),
                                     "note": "the same flow with ww_frontend_params.precise=0 (fp32 butterflies instead of the "
                                             "reference's float64 STFT): FA counts and FRR array against the default profile's"},
           "per_clip_variant": {"note": "round 2's stand-in: every clip evaluated on its own (ring reset, 0.5 s of zeros each side), ALL "
                                        "negatives concatenated as posteriors, + one end-padded window per clip (evaluate_tf_lite_opts.py)",
                                "seconds_host_pcm_in_to_curves_out": el_pc, "median_seconds": el_pc_med, "windows": pc["windows"], "negative_hours": pc["hours"],
                                "audio_frames_per_s": sum((len(c) + 16000) // 160 for c in clips) / el_pc,
                                "frr_at_0.5_fa_per_hour": pc["frr_at_0.5_fa_per_hour"], "fa_count_at_threshold_0.5": int(pc["fa_count"][0]),
                                "one_window_accuracy": pc["one_window_accuracy"], "posterior_checksum": pc["posterior_checksum"]},
           "at_scale": at_scale}
    if with_oracle:
        # the same flow on the C oracle (the checker): ONE framing grid over the padded wake-word files / over the padded
        # stream, windows where StreamPlan (pinned on the literal reference loop in tests/test_host_logic.py) puts them
        from oracle import cpu as ocpu
        from oracle import numpy_ref as NR
        ora = ocpu.CpuOracle(eng.blob)
        ocpu.set_threads(max(1, min(16, granted_cpus())))
        t0 = time.perf_counter()
        pidx = eng.posterior_index
        plan = StreamPlan([len(stream)], eng.window)
        padded = np.zeros(int(plan.padded[0]), np.int16)
        padded[8000:8000 + len(stream)] = stream
        o_neg = ora.slide_forward(ora.logmel(padded, 32768.0, False), 2)[:plan.total, pidx]
        plan = StreamPlan([len(c) for c in wake], eng.window)
        whole = np.zeros(int(plan.padded.sum()), np.int16)
        for k, c in enumerate(wake):
            whole[plan.pos[k] + 8000: plan.pos[k] + 8000 + len(c)] = c
        mel = ora.logmel(whole, 32768.0, False)
        o_pos = np.array([ora.slide_forward(mel[plan.F[k]: plan.F[k] + plan.n_frames[k]], 2)[:plan.n_win[k], pidx].max()
                          for k in range(len(wake))], np.float32)
        of, oa, oc, _ = NR.far_frr(o_pos, o_neg, len(wake), r["hours"])
        # the stand-in's joined stream never crosses 0.5 after smoothing (fa_count_at_threshold_0.5 = 0: FRR @ 0.5 FA/h is FRR at
        # t = 0.5).  The same sweep from 0.02 up, where rising edges exist, on the same posteriors: the device's smoothing + edge
        # counting against the reference's loop (oracle/numpy_ref.py) on the oracle's posteriors
        from wwhip.evaluate import far_frr as gpu_far_frr
        low = np.arange(0.02, 0.5, 0.005)
        _, lf, la, lc = gpu_far_frr(r["positives"], r["negatives"], len(wake), r["hours"], low, engine=eng)
        olf, ola, olc, _ = NR.far_frr(o_pos, o_neg, len(wake), r["hours"], low)
        res["sweep_below_0.5"] = {"thresholds": "arange(0.02, 0.5, 0.005)", "fa_count_max": int(lc.max()), "fa_count_sum": int(lc.sum()),
                                  "thresholds_with_false_accepts": int((lc > 0).sum()),
                                  "frr_at_0.5_fa_per_hour": frr_at_fa(lf, la, 0.5),
                                  "fa_counts_identical_to_oracle": bool(np.array_equal(lc, olc)),
                                  "frr_identical_to_oracle": bool(np.array_equal(lf, olf))}
        res["oracle"] = {"frr_at_0.5_fa_per_hour": frr_at_fa(of, oa, 0.5), "fa_count_at_threshold_0.5": int(oc[0]),
                         "fa_counts_identical": bool(np.array_equal(oc, r["fa_count"])),
                         "frr_identical": bool(np.array_equal(of, r["frr"])),
                         "max_abs_posterior_diff": float(max(np.abs(o_neg - r["negatives"]).max(), np.abs(o_pos - r["positives"]).max())),
                         "seconds": time.perf_counter() - t0,
                         "kind": "oracle/ww_oracle.c + oracle/numpy_ref.py (C restatement, NOT TFLite)"}
    return res


def summary_of(line):
    """Every BASELINE config's figures in one compact object (<= 1.5 KB) at the END of the JSON line: the driver's record keeps
    the last 8 KB of a ~60 KB line, so whatever is to be read from BENCH_rNN.json has to stand here.  Nothing is measured in this
    function; every number is copied from the line (r3 = three significant digits are enough for a record)."""
    def r(v, nd=4):
        return None if v is None else float(f"{float(v):.{nd}g}")

    def g(d, *path):
        for k in path:
            if not isinstance(d, dict) or d.get(k) is None:
                return None
            d = d[k]
        return d
    tr = line.get("timed_regions") or {}
    roof = line.get("roofline") or {}
    s = {"cfg2_crnn": {"value": r(line["value"]), "ms_per_step": r(line["ms_per_step"]), "regions": tr.get("n"),
                       "region_ms_p10_p50_p90": [r(tr.get("p10_ms")), r(tr.get("median_ms")), r(tr.get("p90_ms"))],
                       "timed_s": r(tr.get("timed_seconds"), 3), "kernel": roof.get("kernel"), "kernel_us": r(roof.get("kernel_avg_us")),
                       "frac": r(roof.get("frac"), 3), "single_stream_value": r(g(line, "single_stream", "value")),
                       "fast_profile_value": r(g(line, "fast_profile", "value"))}}
    wv = line.get("wavenet")
    if wv:
        s["cfg3_wavenet"] = {"bf16x3_value": r(g(wv, "bf16x3", "value")), "bf16x3_kernel_us": r(g(wv, "bf16x3", "roofline", "kernel_avg_us")),
                             "bf16x3_frac": r(g(wv, "bf16x3", "roofline", "frac"), 3), "fp32_value": r(g(wv, "fp32_mfma_parity_mode", "value"))}
    st = line.get("streaming")
    if st:
        s["cfg5_tick_ms_p50_p99"] = {k: {"step": [r(v["p50_ms"]), r(v["p99_ms"])],
                                         "pipeline": [r(g(v, "pipeline", "p50_ms")), r(g(v, "pipeline", "p99_ms"))],
                                         "pipeline_over_tick_us": r(g(v, "pipeline", "over_tick_us"), 3)}
                                     for k, v in st.items() if isinstance(v, dict) and "p50_ms" in v}
        s["cfg5_streams_per_gpu"] = st.get("streams_per_gpu")
    ev = line.get("eval_testset")
    if ev:
        sc = ev.get("at_scale") or {}
        s["cfg4"] = {"standin_seconds": r(ev.get("seconds_host_pcm_in_to_curves_out")), "seconds": r(sc.get("seconds_host_pcm_in_to_curves_out")),
                     "serial_ms": r(sc.get("serial_ms")), "one_rank_of_8_s": r(g(sc, "one_rank_of_8_measured", "seconds")),
                     "eff_8_ranks": r(g(sc, "one_rank_of_8_measured", "efficiency_8_ranks"), 3),
                     "x16_seconds": r(g(sc, "at_scale_x16", "seconds_host_pcm_in_to_curves_out")),
                     "x16_one_rank_of_8_s": r(g(sc, "at_scale_x16", "one_rank_of_8_measured", "seconds")),
                     "x16_eff_8_ranks": r(g(sc, "at_scale_x16", "one_rank_of_8_measured", "efficiency_8_ranks"), 3)}
        s["frr_at_0.5_fa_per_hour"] = r(ev.get("frr_at_0.5_fa_per_hour"), 6)
        s["frr_oracle"] = r(g(ev, "oracle", "frr_at_0.5_fa_per_hour"), 6)
        s["fa_frr_identical_to_oracle"] = [g(ev, "oracle", "fa_counts_identical"), g(ev, "oracle", "frr_identical")]
    cb = line.get("cpu_baseline")
    if cb:
        s["cpu_baseline"] = {"value": r(cb.get("value")), "cores": cb.get("cores"), "one_thread": r(g(cb, "one_thread", "value")), "kind": cb.get("kind")}
    return s


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--repeats", type=int, default=0, help="timed regions of K steps each (median reported); 0 = auto: "
                    "enough regions for >= 1 s of timed work, between 5 and 2001")
    ap.add_argument("--model", choices=["crnn", "wavenet"], default="crnn")
    ap.add_argument("--clips", type=int, default=256)
    ap.add_argument("--rotate", type=int, default=24, help="distinct resident input batches to rotate over")
    ap.add_argument("--pipeline", type=int, default=4,
                    help="independent contexts (HIP streams) the steps are dealt to round-robin; batches are "
                         "independent, so consecutive steps may overlap on the GPU")
    ap.add_argument("--fast-frontend", action="store_true", help="fp32 FFT instead of the reference's fp64")
    ap.add_argument("--precision", choices=["auto", "fp32", "bf16x3"], default="auto",
                    help="model contractions: fp32 MFMA, or (Wavenet only) three bf16 MFMAs on split operands with "
                         "fp32 accumulate; auto = fp32 for CRNN (BASELINE cfg 2), bf16x3 for Wavenet (cfg 3)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="headline job only: skip the wavenet / streaming / eval legs")
    ap.add_argument("--stream-ticks", type=int, default=10000)
    ap.add_argument("--eval-clips", type=int, default=2048)
    ap.add_argument("--eval-scale", type=int, default=2529, help="wake-word clips of the at-scale evaluation leg (hey-snips test "
                    "split: 2,529; as many other clips are joined into the negative stream); 0 = skip")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args.gpus))

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    cpu_set = None
    if world > 1 and hasattr(os, "sched_setaffinity"):
        # one node, one rank per GPU: every rank on its own slice of the CPUs this job may use, BEFORE anything touches the
        # GPU (the tick loop of the streaming leg is host-paced: ranks that migrate over each other's cores show up as p99)
        cpus = sorted(os.sched_getaffinity(0))
        lw = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
        per = len(cpus) // max(lw, 1)
        if per >= 1:
            cpu_set = cpus[(local_rank % lw) * per:(local_rank % lw + 1) * per]
            os.sched_setaffinity(0, cpu_set)
    import numpy as np
    import torch
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X; no CPU fallback exists for the hot path")
    # WW_BENCH_BACKEND=gloo is a rehearsal mode for a one-GPU box (ranks share GPU 0, the gather
    # goes through host memory); the driver's multi-GPU runs use RCCL ("nccl"), one GPU per rank.
    backend = os.environ.get("WW_BENCH_BACKEND", "nccl")
    n_dev = torch.cuda.device_count()
    if backend == "nccl" and world > n_dev:
        raise SystemExit(f"--gpus {world} but only {n_dev} GPU(s) visible (WW_BENCH_BACKEND=gloo rehearses on fewer)")
    if backend != "nccl":
        local_rank = local_rank % max(n_dev, 1)
    torch.cuda.set_device(local_rank)
    dist = None
    # WW_BENCH_FORCE_DIST=1: go through the process-group code (barriers, the posterior all_gather, MAX-reduced times) even
    # at world size 1 - lets a one-GPU box exercise the RCCL path the multi-GPU runs take
    if world > 1 or os.environ.get("WW_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29671")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    comm_dev = "cuda" if backend == "nccl" else "cpu"

    from wwhip.engine import Engine, frontend_params
    from wwhip import _lib
    P = max(1, args.pipeline)
    K = args.steps
    R = max(1, args.rotate)
    W = max(args.warmup, P)  # every context once: workspaces sized before the first timed region
    ctxs = [_lib.Context(local_rank) for _ in range(P)]
    assets = os.path.join(PKG, "assets", "tf_lite_models")
    fp = frontend_params(32767.0, True, 0.0, 160, not args.fast_frontend)

    rng = np.random.default_rng(1000 + rank)
    pcm0 = synth_pcm(rng, args.clips)
    d_pcm = [torch.from_numpy(pcm0).cuda()]
    for r in range(1, R):  # cheap decorrelated variants: roll + sign flip, generated on the device
        d_pcm.append(torch.roll(d_pcm[0], shifts=37 * r, dims=1) * (1 if r % 2 == 0 else -1))
    torch.cuda.synchronize()

    def run_model(model, precision, want_profile):
        engs = [Engine(os.path.join(assets, "CRNN" if model == "crnn" else "Wavenet"), device=local_rank, ctx=c,
                       precision=precision) for c in ctxs]
        job = Job(torch, engs, ctxs, d_pcm, args.clips, K, dist, comm_dev, world)
        job.warm(W, fp)
        n_rep = args.repeats
        if n_rep <= 0:
            probe = sorted(job.region(fp)[0] for _ in range(3))[1]  # (the median of three: one slow probe region must not halve the count)
            # >= 1 s of timed headline work (round 5 timed 41 regions of 1 ms under the driver's --steps 20: 39 ms in all)
            n_rep = int(min(2001, max(5, 1.1 / max(probe, 1e-6)))) | 1
        stats, med = job.regions(n_rep, fp)
        res = {"engs": engs, "job": job, "stats": stats, "median_s": med}
        res["posts"] = job.d_all[:min(K, R)].cpu().numpy()
        if want_profile:
            job.warm(1, fp)
            s1, m1 = job.regions(max(3, n_rep // 2) | 1, fp, only0=True)
            res["single"] = {"value": throughput(world, K, args.clips, m1), "unit": "audio frames/s",
                             "ms_per_step": m1 / K * 1e3, "timed_regions": s1,
                             "note": "same job, every step on ONE context / HIP stream (--pipeline 1): front end -> "
                                     "model kernels of a batch strictly in sequence"}
            res["prof"] = profile_pass(job, ctxs[0], K, fp)
        return res

    precision = args.precision if args.precision != "auto" else ("fp32" if args.model == "crnn" else "bf16x3")
    if args.model == "crnn":
        precision = "fp32"
    head = run_model(args.model, precision, True)
    eng = head["engs"][0]

    # ---- same regions with the fp32-FFT front end (precise=0), reported as a side figure:
    # posteriors stay within 5e-6 of the fp64-FFT path (tools/f32_error.py), log-mel within 1.4e-4
    alt = None
    if not args.fast_frontend:
        fp_fast = frontend_params(32767.0, True, 0.0, 160, False)
        head["job"].warm(P, fp_fast)
        s, m = head["job"].regions(max(3, head["stats"]["n"] // 2) | 1, fp_fast)
        alt = {"value": throughput(world, K, args.clips, m), "unit": "audio frames/s", "ms_per_step": m / K * 1e3,
               "timed_regions": s, "note": "same job with ww_frontend_params.precise=0 (fp32 butterflies)"}

    # ---- CRNN only: the same regions with conv + projection on split-bf16 MFMA (ww_model_set_precision(BF16X3)); fp32 stays
    # the headline (BASELINE configs[1] is an fp32 configuration), this is what the bf16 matrix pipe buys at <= 3e-5 on posteriors
    alt_bf16 = fast_profile = None
    if args.model == "crnn" and precision == "fp32" and not args.fast_frontend:
        for e in head["engs"]:
            e.set_precision("bf16x3")
        head["job"].warm(P, fp)
        s, m = head["job"].regions(max(3, head["stats"]["n"] // 2) | 1, fp)
        posts_b = head["job"].d_all[:min(K, R)].cpu().numpy()
        alt_bf16 = {"value": throughput(world, K, args.clips, m), "unit": "audio frames/s", "ms_per_step": m / K * 1e3,
                    "timed_regions": s, "max_abs_posterior_diff_vs_fp32": float(np.abs(posts_b - head["posts"]).max()),
                    "note": "same job, CRNN conv + layer-1 projection as split-bf16 MFMA products (3 per product, fp32 "
                            "accumulate); recurrences, head and the fp64-FFT front end unchanged"}
        pb = profile_pass(head["job"], ctxs[0], K, fp, min_launches=100)
        alt_bf16["all_kernels_avg_us"] = {k: round(v["total_ms"] / max(v["calls"], 1) * 1e3, 3) for k, v in pb.items()}
        # ---- the documented fast profile: fp32-FFT front end + split-bf16 conv / projection together.  Not the headline
        # (BASELINE configs[1] is fp32 and the reference's STFT is float64); licensed by tests/test_gpu_bench_eval.py::
        # test_fp32_fft_front_end_leaves_far_frr_untouched (FA counts and FRR arrays identical at cfg-1 scale) and the
        # split-bf16 bound (posteriors within 5e-5 of fp32)
        head["job"].warm(P, fp_fast)
        s, m = head["job"].regions(max(3, head["stats"]["n"] // 2) | 1, fp_fast)
        posts_fb = head["job"].d_all[:min(K, R)].cpu().numpy()
        fast_profile = {"value": throughput(world, K, args.clips, m), "unit": "audio frames/s", "ms_per_step": m / K * 1e3,
                        "timed_regions": s, "max_abs_posterior_diff_vs_default": float(np.abs(posts_fb - head["posts"]).max()),
                        "profile": "ww_frontend_params.precise=0 (fp32 FFT) + ww_model_set_precision(BF16X3) (CRNN conv + layer-1 "
                                   "projection as split-bf16 MFMA); recurrences and head fp32",
                        "far_frr": "identical to the default profile at cfg-1 scale: eval_testset.fast_profile_fp32_fft and "
                                   "tests/test_gpu_bench_eval.py (the evaluation flows use the fp32 sliding form in either "
                                   "precision mode, so the front end is the only difference there)"}
        for e in head["engs"]:
            e.set_precision("fp32")

    def close(res):
        for e in res["engs"]:
            e.close()

    extra = {}
    if not args.no_extra:
        # ---- BASELINE configs[2] (or configs[1] when the headline is the Wavenet): the other model, same regions
        if args.model == "crnn":
            wv = run_model("wavenet", "bf16x3", True)
            for e in wv["engs"]:
                e.set_precision("fp32")
            wv["job"].warm(P, fp)
            s32, m32 = wv["job"].regions(max(3, wv["stats"]["n"] // 2) | 1, fp)
            posts32 = wv["job"].d_all[:min(K, R)].cpu().numpy()
            weng = wv["engs"][0]
            extra["wavenet"] = {
                "workload": f"Wavenet dilated-conv encode+detect, batch={args.clips}x1.5 s clips per GPU, same PCM, "
                            f"{weng.window}x40 window (BASELINE configs[2])",
                "bf16x3": {"value": throughput(world, K, args.clips, wv["median_s"]), "unit": "audio frames/s",
                           "ms_per_step": wv["median_s"] / K * 1e3, "timed_regions": wv["stats"],
                           "dtype": "bf16x3 (split-bf16 MFMA products, f32 accumulate)",
                           "single_stream": wv["single"], "roofline": roofline_of(weng, wv["prof"], args.clips, args.fast_frontend)},
                "fp32_mfma_parity_mode": {"value": throughput(world, K, args.clips, m32), "unit": "audio frames/s",
                                          "ms_per_step": m32 / K * 1e3, "timed_regions": s32,
                                          "max_abs_posterior_diff_vs_bf16x3": float(np.abs(posts32 - wv["posts"]).max())}}
            close(wv)
        # ---- BASELINE configs[4]: streaming
        extra["streaming"] = stream_leg(torch, np, dist, comm_dev, local_rank, 128, args.stream_ticks)
        # ---- BASELINE configs[0] stand-in / configs[3]: sharded evaluation, FRR @ 0.5 FA/h
        extra["eval_testset"] = eval_leg(torch, np, dist, comm_dev, rank, world, local_rank, args.eval_clips,
                                         with_oracle=(world == 1 and not args.no_cpu_baseline), n_scale=args.eval_scale)

    if rank == 0:
        med = head["median_s"]
        roof = roofline_of(eng, head["prof"], args.clips, args.fast_frontend, step_s=med / K)
        line = {
            "metric": "audio frames/sec (16 kHz, 40-mel, 10 ms hop) + FRR@0.5 FA/h",
            "value": throughput(world, K, args.clips, med),
            "unit": "audio frames/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": med / K * 1e3,
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
                "resident_input_bytes": R * args.clips * SAMPLES * 2,
                "pipelined_contexts": P,
                "weights": "reference tf_lite_models (shipped fp32 weights)",
                "parallelism": f"utterance-sharded x{world}, posterior all_gather once per timed region" if world > 1 else "single GPU",
                "cpus_per_rank": len(cpu_set) if cpu_set else None,
            },
            "timed_regions": dict(head["stats"], note="each region = exactly `steps` steps; barrier+synchronize in front, the clock "
                                  "stops when the posterior gather (N > 1) has completed on the rank, MAX over ranks; value and "
                                  "ms_per_step are the median region's; collective_ms = the gather alone (warm-up, median of 4)"),
            "roofline": roof,
            "posterior_checksum": float(np.sum(head["posts"][0], dtype=np.float64)),
            "single_stream": head["single"],
            "alt_fp32_fft_frontend": alt,
            "alt_crnn_split_bf16": alt_bf16,
            "fast_profile": fast_profile,
        }
        if world > 1 and backend != "nccl":
            line["config"]["rehearsal"] = f"{backend} backend, ranks share {n_dev} physical GPU(s): not a scaling measurement"
        line.update(extra)
        if extra.get("eval_testset"):
            line["frr_at_0.5_fa_per_hour"] = extra["eval_testset"]["frr_at_0.5_fa_per_hour"]
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(eng, pcm0, float(os.environ.get("WW_BENCH_CPU_SECONDS", "10")))
        elif not args.no_cpu_baseline:
            line["cpu_baseline"] = None
        line["summary"] = summary_of(line)  # LAST key: the driver's record keeps the last 8 KB of this line
        print(json.dumps(line), flush=True)
    close(head)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
