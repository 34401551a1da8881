"""bench.py's contract (one JSON line, all BASELINE configs, self-launched ranks) and the sharded
test-set evaluation (SURVEY 8d cfg 1 at full size, cfg 4 as 1 rank == 2 ranks) - on the MI355X.

Every bench / multi-rank run is a CHILD process: nothing of this test process's GPU state is shared,
and the 2-rank runs are gloo rehearsals in which both ranks use GPU 0 (one-GPU box)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOL = 1e-4  # north_star: posteriors within 1e-4 of the reference arithmetic (here: of the oracle)


def _bench(args, env=None, timeout=900):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args, capture_output=True, text=True,
                       timeout=timeout, env=dict(os.environ, WW_BENCH_CPU_SECONDS="1", **(env or {})))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    return json.loads(lines[0])


def test_bench_line_honours_the_contract():
    """`python bench.py --steps K --warmup W` prints ONE JSON line with the driver's fields, the roofline of the
    dominant kernel, the CPU baseline, and the legs for BASELINE configs[0], [2], [3]-as-1-rank and [4]."""
    d = _bench(["--steps", "40", "--warmup", "4", "--stream-ticks", "300", "--eval-clips", "256"])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "timed_regions", "single_stream",
                "wavenet", "streaming", "eval_testset", "frr_at_0.5_fa_per_hour"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 40 and d["warmup"] == 4 and d["higher_is_better"] is True
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["data"] == "synthetic"
    assert "workload" in d["config"] and "model" not in d["config"]
    # the input rotation does not shrink with K: more resident PCM than the 256 MiB Infinity Cache
    assert d["config"]["resident_input_batches_rotated"] >= 24 and d["config"]["resident_input_bytes"] > 256 << 20
    frames = 40 * 256 * 150  # K steps x 256 clips x 10 ms hops of a 1.5 s clip
    assert abs(d["value"] - frames / (d["ms_per_step"] * 40 * 1e-3)) / d["value"] < 1e-6
    tr = d["timed_regions"]
    assert tr["n"] >= 5 and tr["min_ms"] <= tr["p10_ms"] <= tr["median_ms"] <= tr["p90_ms"] <= tr["max_ms"]
    assert tr["timed_seconds"] >= 0.8  # (round 6) the headline rests on >= 1 s of timed work, not on 41 regions of 1 ms
    assert abs(tr["median_ms"] - d["ms_per_step"] * 40) < 1e-6 * tr["median_ms"] + 1e-9
    roof = d["roofline"]
    assert roof["bound"] in ("hbm", "mfma") and roof["unit"] in ("GB/s", "TFLOP/s")
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-9 and 0 < roof["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["sample"]
    assert d["value"] > 10 * cb["value"]  # north_star: >= 10x the CPU path at 1 GPU
    assert 0 < d["single_stream"]["value"] <= d["value"] * 1.05
    wv = d["wavenet"]
    assert wv["bf16x3"]["value"] > 0 and wv["fp32_mfma_parity_mode"]["value"] > 0
    assert wv["fp32_mfma_parity_mode"]["max_abs_posterior_diff_vs_bf16x3"] < TOL
    st = d["streaming"]
    assert st["streams_per_gpu"] == 128 and st["ticks"] == 300
    for k in ("crnn", "wavenet_bf16x3", "wavenet"):
        assert 0 < st[k]["p50_ms"] <= st[k]["p99_ms"] < 20.0  # a tick is 20 ms of audio
        assert st[k]["posteriors_per_tick"] == 256.0
        # (round 5) where a tick's host time goes: the C entry point's phases + the Python wrapper; a tick is ONE launch
        hp = st[k]["host_phases_us"]
        assert set(hp) == {"plan", "frames_in", "launch_1", "launch_2", "wait", "copy_out"} and hp["launch_2"] < 0.5 < hp["launch_1"]
        assert abs(sum(hp.values()) + st[k]["python_wrapper_us"] - st[k]["mean_ms"] * 1e3) < 1e-2
        # (round 6) the same tick at the plugin surface - VadBank -> WakewordBank -> ActivationTimeoutBank on a ContextBank -
        # costs a few microseconds more than the bare tick, not six times as much
        pl = st[k]["pipeline"]
        assert 0 < pl["p50_ms"] <= pl["p99_ms"] < 20.0 and pl["over_tick_us"] < 15.0, pl
    ev = d["eval_testset"]
    assert ev["world_size"] == 1 and 0.0 <= ev["frr_at_0.5_fa_per_hour"] <= 1.0
    assert ev["oracle"]["fa_counts_identical"] and ev["oracle"]["frr_identical"] and ev["oracle"]["max_abs_posterior_diff"] < TOL
    assert d["frr_at_0.5_fa_per_hour"] == ev["frr_at_0.5_fa_per_hour"] == ev["oracle"]["frr_at_0.5_fa_per_hour"]
    # the stand-in is FA-free at the reference's thresholds: the sweep below 0.5, where rising edges exist, against the reference's loop
    lo = ev["sweep_below_0.5"]
    assert lo["fa_count_max"] > 0 and lo["fa_counts_identical_to_oracle"] and lo["frr_identical_to_oracle"]
    # (round 5) configs[3] at the size it names says what 8 GPUs can give: the phases every rank repeats, the prediction
    # from this pass's phases, one rank's share of a world of 8 measured alone, and the same over 16 x the clips
    sc = ev["at_scale"]
    assert 0 < sc["serial_ms"] < 3.0 and sc["median_seconds"] >= sc["seconds_host_pcm_in_to_curves_out"]
    assert sc["predicted_seconds"]["8"] < sc["predicted_seconds"]["2"] < sc["seconds_host_pcm_in_to_curves_out"]
    assert 1.0 < sc["one_rank_of_8_measured"]["speedup_vs_one_rank"] <= 8.0
    x16 = sc["at_scale_x16"]
    assert x16["windows"] > 15 * sc["windows"] and x16["one_rank_of_8_measured"]["speedup_vs_one_rank"] > sc["one_rank_of_8_measured"]["speedup_vs_one_rank"] * 0.9
    assert ev["median_seconds"] >= ev["seconds_host_pcm_in_to_curves_out"]
    # (round 6) every config's figures once more in a compact object at the END of the line (the driver keeps its last 8 KB)
    assert list(d)[-1] == "summary" and len(json.dumps(d["summary"])) <= 1536, len(json.dumps(d["summary"]))
    sm = d["summary"]
    assert set(sm) >= {"cfg2_crnn", "cfg3_wavenet", "cfg5_tick_ms_p50_p99", "cfg4", "frr_at_0.5_fa_per_hour", "frr_oracle", "cpu_baseline"}
    assert abs(sm["cfg2_crnn"]["value"] - d["value"]) < 1e-3 * d["value"]
    assert set(sm["cfg5_tick_ms_p50_p99"]) == {"crnn", "wavenet_bf16x3", "wavenet"}
    assert all(v is not None for v in sm["cfg4"].values()) and sm["frr_at_0.5_fa_per_hour"] == sm["frr_oracle"]


def test_bench_starts_its_own_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE: the parent (no torch, no GPU) starts two child ranks and relays
    rank 0's line.  One-GPU box: gloo rehearsal, both ranks on GPU 0 - the line says so."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12", "--warmup", "4",
                        "--stream-ticks", "100", "--eval-clips", "128", "--eval-scale", "40"], capture_output=True, text=True, timeout=900,
                       env=dict(env, WW_BENCH_BACKEND="gloo", WW_BENCH_CPU_SECONDS="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 12 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert abs(d["value"] - 2 * 12 * 256 * 150 / (d["ms_per_step"] * 12 * 1e-3)) / d["value"] < 1e-6
    assert "rehearsal" in d["config"] and "x2" in d["config"]["parallelism"]
    assert d["eval_testset"]["world_size"] == 2 and d["streaming"]["crnn"]["p50_ms"] > 0
    assert d["streaming"]["crnn"]["p99.9_ms"] >= d["streaming"]["crnn"]["p99_ms"] >= d["streaming"]["crnn"]["p50_ms"]
    tr = d["timed_regions"]     # the N > 1 line says what the collective costs and how far the ranks are apart
    assert tr["collective_ms"] > 0 and 0 < tr["per_rank_median_ms"]["min"] <= tr["per_rank_median_ms"]["max"] <= tr["max_ms"] * 1.001
    assert d["config"]["cpus_per_rank"] >= 1


def test_bench_under_torch_distributed_run():
    """The driver's launch form for N > 1: `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1
    --master-port P bench.py --gpus N --steps K --warmup W` - the ranks come from the launcher (RANK / WORLD_SIZE in the environment),
    bench.py must not start its own.  One-GPU box: gloo rehearsal.  The evaluation leg's numbers equal the one-rank line's."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29668", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "12",
                        "--warmup", "4", "--stream-ticks", "100", "--eval-clips", "256", "--eval-scale", "60"], capture_output=True, text=True, timeout=900,
                       env=dict(env, WW_BENCH_BACKEND="gloo", WW_BENCH_CPU_SECONDS="1"))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 12 and d["eval_testset"]["world_size"] == 2 and d["cpu_baseline"] is None
    tr = d["timed_regions"]
    assert tr["collective_ms"] > 0 and 0 < tr["per_rank_median_ms"]["min"] <= tr["per_rank_median_ms"]["max"]
    one = _bench(["--steps", "12", "--warmup", "4", "--stream-ticks", "100", "--eval-clips", "256", "--eval-scale", "60", "--no-cpu-baseline"])
    for k in ("frr_at_0.5_fa_per_hour", "fa_count_at_threshold_0.5", "posterior_checksum", "windows", "negative_hours"):
        assert d["eval_testset"][k] == one["eval_testset"][k], k
    a2, a1 = d["eval_testset"]["at_scale"], one["eval_testset"]["at_scale"]    # the at-scale leg (reduced here): 2 ranks = 1 rank
    for k in ("frr_at_0.5_fa_per_hour", "fa_count_at_threshold_0.5", "posterior_checksum", "windows"):
        assert a2[k] == a1[k], k
    assert a2["world_size"] == 2 and 0.0 < a2["host_share"] < 1.0 and a2["device_ms"] > 0
    assert {"plan", "slicing", "submit", "upload_wait", "device_wall", "gather", "sweep"} <= set(a2["host_phases_ms"])
    assert d["eval_testset"]["per_clip_variant"]["posterior_checksum"] == one["eval_testset"]["per_clip_variant"]["posterior_checksum"]


@pytest.fixture(scope="module")
def testset_one_rank(assets):
    from wwhip.evaluate import synth_testset, evaluate_testset_sharded
    from wwhip.models import engine_for
    eng = engine_for(os.path.join(assets, "CRNN_softmax"))
    clips, labels = synth_testset(2048)
    return eng, clips, labels, evaluate_testset_sharded(eng, clips, labels)


def test_cfg4_sharded_evaluation_identical_to_one_rank(testset_one_rank, tmp_path):
    """BASELINE configs[3] / SURVEY 8(d) cfg 4 at full size (2,048 clips, 118,643 windows): utterances dealt over
    2 ranks + posterior gather gives the FRR array, the FA counts and every posterior bit-identical to 1 rank."""
    _, _, _, one = testset_one_rank
    dump = tmp_path / "two_ranks.npz"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", "29653", os.path.join(ROOT, "tools", "eval_testset.py"),
                        "--backend", "gloo", "--dump", str(dump)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["world_size"] == 2 and line["clips"] == 2048
    two = np.load(dump)
    np.testing.assert_array_equal(two["frr"], one["frr"])
    np.testing.assert_array_equal(two["fa_count"], one["fa_count"])
    np.testing.assert_array_equal(two["sliding"], one["sliding"])
    np.testing.assert_array_equal(two["one"], one["one_window_posteriors"])
    assert float(two["checksum"]) == one["posterior_checksum"] == line["posterior_checksum"]
    assert line["frr_at_0.5_fa_per_hour"] == one["frr_at_0.5_fa_per_hour"]


def test_cfg1_full_size_vs_oracle(testset_one_rank):
    """BASELINE configs[0] stand-in at full size: the GPU's FRR @ 0.5 FA/h, FRR array and FA counts equal the CPU
    oracle's flow (C restatement for the posteriors, the NumPy restatement of plot_FRR_FAR for the sweep)."""
    from oracle import cpu as ocpu
    from oracle import numpy_ref as NR
    from wwhip.evaluate import frr_at_fa
    eng, clips, labels, one = testset_one_rank
    ora = ocpu.CpuOracle(eng.blob)
    ocpu.set_threads(max(1, min(16, os.cpu_count() or 1)))
    lab, offs, pidx = labels.astype(bool), one["sliding_offsets"], eng.posterior_index
    z = np.zeros(8000, np.int16)
    o_pos, o_neg, o_one = [], [], []
    for i, c in enumerate(clips):
        p = ora.slide_forward(ora.logmel(np.concatenate((z, c, z))), 2)[:, pidx]
        assert len(p) == offs[i + 1] - offs[i]
        assert np.abs(p - one["sliding"][offs[i]:offs[i + 1]]).max() < TOL
        (o_pos if lab[i] else o_neg).append(p.max() if lab[i] else p)
        if i % 16 == 0:  # the a17 flow (one end-padded window per clip) on a sample
            mel = ora.logmel(c)
            w = np.zeros((eng.window, 40), np.float32)
            w[:min(len(mel), eng.window)] = mel[:eng.window]
            assert abs(ora.forward(w)[0, pidx] - one["one_window_posteriors"][i]) < TOL
    wf, wa, wc, _ = NR.far_frr(np.array(o_pos, np.float32), np.concatenate(o_neg), int(lab.sum()), one["hours"])
    np.testing.assert_array_equal(one["fa_count"], wc)
    np.testing.assert_array_equal(one["frr"], wf)
    assert one["frr_at_0.5_fa_per_hour"] == frr_at_fa(wf, wa, 0.5)


@pytest.fixture(scope="module")
def reference_flow_one_rank(assets):
    from wwhip.evaluate import synth_testset, evaluate_reference_flow_sharded
    from wwhip.models import engine_for
    eng = engine_for(os.path.join(assets, "CRNN_softmax"))
    clips, labels = synth_testset(2048)
    return eng, clips, labels, evaluate_reference_flow_sharded(eng, clips, labels)


def _write_wav(path, pcm):
    import wave
    with wave.open(str(path), "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(np.asarray(pcm, np.int16).tobytes())


@pytest.mark.parametrize("world", [2, 3])
def test_reference_flow_sharded_identical_for_any_rank_count(reference_flow_one_rank, tmp_path, world):
    """SURVEY 8(e), the reference's own false-accept flow (utils/evaluate_models.py main()): the first num_wakewords
    negative clips joined by 100 ms of silence into ONE stream, evaluated continuously and cut into `world` contiguous
    posterior ranges (each rank front-ends only its samples + the T-2-frame overlap), positives utterance-sharded with
    the never-reset ring's carry (quirk C2): every posterior, the FA counts and the FRR array are bit-identical to 1 rank."""
    _, _, _, one = reference_flow_one_rank
    dump = tmp_path / f"ranks{world}.npz"
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
                        "--master-addr", "127.0.0.1", "--master-port", str(29660 + world), os.path.join(ROOT, "tools", "eval_testset.py"),
                        "--flow", "reference", "--backend", "gloo", "--dump", str(dump)], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert line["world_size"] == world and line["flow"] == "reference"
    got = np.load(dump)
    np.testing.assert_array_equal(got["negatives"], one["negatives"])
    np.testing.assert_array_equal(got["positives"], one["positives"])
    np.testing.assert_array_equal(got["fa_count"], one["fa_count"])
    np.testing.assert_array_equal(got["frr"], one["frr"])
    assert float(got["checksum"]) == one["posterior_checksum"] == line["posterior_checksum"]
    assert line["frr_at_0.5_fa_per_hour"] == one["frr_at_0.5_fa_per_hour"]


@pytest.mark.timeout(600)
def test_reference_flow_does_not_depend_on_the_staging_chunks(reference_flow_one_rank, monkeypatch):
    """The sharded flow sends a rank's share to the GPU in chunks through a staging pipeline (wwhip.evaluate._run_jobs: a
    host thread stages and uploads chunk i + 1 while chunk i computes).  Where the cuts fall must not show: with chunks of
    about 20 s of audio (dozens of chunks per leg, the three page-locked slots reused many times, negative-stream cuts inside
    clips and inside gaps, tail kernels of every size class) every posterior, FA count and FRR value equals the default
    run's, bit for bit; and a clip that is not int16 sends its chunk through the float32 path without upsetting the rest."""
    from wwhip import evaluate as E
    eng, clips, labels, one = reference_flow_one_rank
    for chunk in (320_000, 1_000_000):
        monkeypatch.setattr(E, "_CHUNK_SAMPLES", chunk)
        tm = {}
        got = E.evaluate_reference_flow_sharded(eng, clips, labels, timing=tm)
        assert tm["chunks"] > (10 if chunk > 500_000 else 40) and tm["device_ms"] > 0
        for key in ("negatives", "positives", "fa_count", "frr"):
            np.testing.assert_array_equal(got[key], one[key])
        assert got["posterior_checksum"] == one["posterior_checksum"] and got["windows"] == one["windows"]
    # (round 6) consecutive chunks go to alternating lanes - the same model on contexts (HIP streams) of their own, so that two
    # chunks' kernels run beside each other: one, two or three lanes, the same bits; the kernel table adds up over the lanes
    monkeypatch.setattr(E, "_CHUNK_SAMPLES", 1_000_000)
    for lanes in (1, 2, 3):
        monkeypatch.setattr(E, "_EVAL_LANES", lanes)
        tm = {}
        got = E.evaluate_reference_flow_sharded(eng, clips, labels, timing=tm)
        for key in ("negatives", "positives", "fa_count", "frr"):
            np.testing.assert_array_equal(got[key], one[key])
        assert tm["chunks"] > 10 and tm["device_ms"] > 0 and "crnn_rows_kernel" in tm["kernels_ms"]
    monkeypatch.setattr(E, "_EVAL_LANES", 2)
    assert eng.lane(0) is eng and eng.lane(1) is not eng and eng.lane(1).ctx.stream != eng.ctx.stream and eng.lane(1) is eng.lane(1)
    with eng.options(crnn_tail_mfma=0):
        assert eng.lane(1)._options["crnn_tail_mfma"] == 0   # a lane follows its engine's options at every call
    assert eng.lane(1)._options["crnn_tail_mfma"] == eng._options["crnn_tail_mfma"]
    # get_posterior_sharded over arrays in memory, one of them float32 (librosa's scale): the chunk that holds it (and only
    # that one) takes the float path; int16 / 32768 is exact, so nothing changes
    monkeypatch.setattr(E, "_CHUNK_SAMPLES", 200_000)
    wake = [c for c, l in zip(clips, labels) if l][:24]
    want = E.get_posterior_sharded(eng.model_dir, "CRNN", "false_negatives", wake, 20, 16000, engine=eng, as_array=True)
    np.testing.assert_array_equal(want, one["positives"][:24])
    mixed = list(wake)
    mixed[7] = wake[7].astype(np.float32) / np.float32(32768.0)
    got = E.get_posterior_sharded(eng.model_dir, "CRNN", "false_negatives", mixed, 20, 16000, engine=eng, as_array=True)
    np.testing.assert_array_equal(got, want)
    # a failure while launching (here: made up, at the third chunk, with later chunks already submitted to the uploader)
    # comes out as itself, and the next call finds the pipeline in order
    calls, real = [], E._chunk_forward

    def failing(e, ch, precise, ph):
        calls.append(ch)
        if len(calls) == 3:
            raise RuntimeError("made-up launch failure")
        return real(e, ch, precise, ph)

    monkeypatch.setattr(E, "_chunk_forward", failing)
    with pytest.raises(RuntimeError, match="made-up launch failure"):
        E.get_posterior_sharded(eng.model_dir, "CRNN", "false_negatives", wake, 20, 16000, engine=eng, as_array=True)
    monkeypatch.setattr(E, "_chunk_forward", real)
    np.testing.assert_array_equal(E.get_posterior_sharded(eng.model_dir, "CRNN", "false_negatives", wake, 20, 16000, engine=eng,
                                                          as_array=True), want)


def test_reference_flow_equals_get_posterior_and_oracle(reference_flow_one_rank, tmp_path):
    """The sharded flow's arrays against (a) the drop-in `get_posterior` on the same audio as wav FILES (the world-size-1 case
    of the same implementation, one piece per file instead of arrays in memory): bit-identical; (b) the C oracle over the same stream + the NumPy restatement of plot_FRR_FAR:
    posteriors within 1e-4, FA counts / FRR array / FRR @ 0.5 FA/h identical."""
    from oracle import cpu as ocpu
    from oracle import numpy_ref as NR
    from wwhip.evaluate import get_posterior, join_negatives, frr_at_fa, StreamPlan
    eng, clips, labels, one = reference_flow_one_rank
    lab = labels.astype(bool)
    wake = [c for c, l in zip(clips, lab) if l]
    other = [c for c, l in zip(clips, lab) if not l]
    assert one["num_wakewords"] == len(wake) > 100
    stream = join_negatives(other, len(wake))
    assert len(stream) == sum(len(c) for c in other[:len(wake)]) + 1600 * (len(wake) - 1)
    assert one["hours"] == len(stream) / 16000 / 3600                       # duration_test: the wav's own length
    far = tmp_path / "not_hey_snips_long.wav"
    _write_wav(far, stream)
    neg = np.asarray(get_posterior(eng.model_dir, "CRNN", "false_accepts", [str(far)], 20, 16000), np.float32)
    np.testing.assert_array_equal(neg, one["negatives"])
    files = []
    for i, c in enumerate(wake[:40]):
        files.append(str(tmp_path / f"w{i}.wav"))
        _write_wav(files[-1], c)
    pos = np.asarray(get_posterior(eng.model_dir, "CRNN", "false_negatives", files, 20, 16000), np.float32)
    np.testing.assert_array_equal(pos, one["positives"][:40])
    # (b) the oracle
    ora = ocpu.CpuOracle(eng.blob)
    ocpu.set_threads(max(1, min(16, os.cpu_count() or 1)))
    pidx = eng.posterior_index
    plan = StreamPlan([len(stream)], eng.window)
    padded = np.zeros(int(plan.padded[0]), np.int16)
    padded[8000:8000 + len(stream)] = stream
    o_neg = ora.slide_forward(ora.logmel(padded, 32768.0, False), 2)[:, pidx]
    assert len(o_neg) == len(one["negatives"]) == plan.total
    assert np.abs(o_neg - one["negatives"]).max() < TOL
    plan = StreamPlan([len(c) for c in wake], eng.window)
    whole = np.zeros(int(plan.padded.sum()), np.int16)
    for k, c in enumerate(wake):
        whole[plan.pos[k] + 8000: plan.pos[k] + 8000 + len(c)] = c
    mel = ora.logmel(whole, 32768.0, False)                                  # ONE grid over all files: the never-reset ring
    o_pos = []
    for k in range(len(wake)):
        rows = mel[plan.F[k]: plan.F[k] + plan.n_frames[k]]
        p = ora.slide_forward(rows, 2)[:plan.n_win[k], pidx]                 # windows still pending at end of file are dropped
        assert len(p) == plan.n_win[k]
        o_pos.append(p.max())
    assert np.abs(np.array(o_pos) - one["positives"]).max() < TOL
    wf, wa, wc, _ = NR.far_frr(np.array(o_pos, np.float32), o_neg, len(wake), one["hours"])
    np.testing.assert_array_equal(one["fa_count"], wc)
    np.testing.assert_array_equal(one["frr"], wf)
    assert one["frr_at_0.5_fa_per_hour"] == frr_at_fa(wf, wa, 0.5)
    # this stand-in's joined stream never crosses 0.5 after smoothing (FA counts all zero above), so the sweep over the stream
    # is exercised at thresholds it does cross: the GPU sweep of the GPU posteriors against plot_FRR_FAR's restatement on the
    # same numbers (identical counts: no tolerance involved), smoothing across the former rank cuts included
    from wwhip.evaluate import far_frr
    low = np.arange(0.02, 0.5, 0.01)
    _, frr_l, fa_l, cnt_l = far_frr(one["positives"], one["negatives"], len(wake), one["hours"], thresholds=low, engine=eng)
    rf, ra, rc, _ = NR.far_frr(one["positives"], one["negatives"], len(wake), one["hours"], thresholds=low)
    assert rc.max() > 20 and rc[-1] == 0                         # rising edges are really counted
    np.testing.assert_array_equal(cnt_l, rc)
    np.testing.assert_array_equal(frr_l, rf)
    np.testing.assert_allclose(fa_l, ra, rtol=1e-15)


def test_fp32_fft_front_end_leaves_far_frr_untouched(testset_one_rank, reference_flow_one_rank):
    """Quirk C5 / VERDICT r2 item 3, decided on evidence at BASELINE cfg-1 scale: with ww_frontend_params.precise = 0 (fp32
    butterflies instead of the reference's float64 STFT, utils/tf_lite/filter.py:59-68) the FA counts and the FRR array of
    BOTH evaluation flows (118,643 per-clip windows; the joined negative stream + carried positives) are identical to the
    default profile's - which equal the oracle's (tests above) - and no posterior moves by more than 5e-6.  fp64 stays the
    library default; this is what licenses the documented fast profile in bench.py's line."""
    from wwhip.engine import frontend_params
    from wwhip.evaluate import evaluate_testset_sharded, evaluate_reference_flow_sharded
    eng, clips, labels, one = testset_one_rank
    _, _, _, ref = reference_flow_one_rank
    fast = evaluate_testset_sharded(eng, clips, labels, fp=frontend_params(32767.0, True, 0.0, 160, False))
    np.testing.assert_array_equal(fast["fa_count"], one["fa_count"])
    np.testing.assert_array_equal(fast["frr"], one["frr"])
    d1 = float(np.abs(fast["sliding"] - one["sliding"]).max())
    assert 0.0 < d1 < 5e-6, d1                                   # a different front end (not a no-op), far inside 1e-4
    fast_ref = evaluate_reference_flow_sharded(eng, clips, labels, precise=False)
    np.testing.assert_array_equal(fast_ref["fa_count"], ref["fa_count"])
    np.testing.assert_array_equal(fast_ref["frr"], ref["frr"])
    d2 = float(max(np.abs(fast_ref["negatives"] - ref["negatives"]).max(), np.abs(fast_ref["positives"] - ref["positives"]).max()))
    assert 0.0 < d2 < 5e-6, d2
    assert fast_ref["frr_at_0.5_fa_per_hour"] == ref["frr_at_0.5_fa_per_hour"]


def test_bench_rccl_path_at_world_size_one():
    """The multi-GPU runs use the `nccl` (= RCCL) backend with device tensors in the posterior gather and the timing
    reductions; a one-GPU box can at least run that code at world size 1 (WW_BENCH_FORCE_DIST=1)."""
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "12", "--warmup", "4", "--no-extra",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=600,
                       env=dict(env, WW_BENCH_FORCE_DIST="1", HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert d["n_gpus"] == 1 and d["value"] > 0 and d["timed_regions"]["n"] >= 5


def test_posterior_gather_over_rccl(tmp_path):
    """wwhip.dist.gather_posteriors and gather_values_t with the nccl backend and CUDA payloads (one rank: the collectives still run)."""
    script = tmp_path / "g.py"
    script.write_text(f"""
import os, sys
sys.path[:0] = [{ROOT!r}, os.path.join({ROOT!r}, "wakeword-detection_amd")]
import numpy as np, torch, torch.distributed as dist
from wwhip import dist as D
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
idx = [4, 0, 2]
full = D.gather_posteriors(np.array([0.4, 0.0, 0.2], np.float32), idx, 6, device="cuda")
assert full.tolist() == [0.0, 0.0, np.float32(0.2), 0.0, np.float32(0.4), 0.0], full
# the sharded evaluators' exchange (round 5): values stay on the GPU - device tensor in, one all_gather, device tensors out
vals = torch.arange(1000, dtype=torch.float32, device="cuda") * 0.5
(back,) = D.gather_values_t(vals, [1000], device="cuda")
assert back.is_cuda and back.device == vals.device and torch.equal(back, vals)
(empty,) = D.gather_values_t(vals[:0], [0], device=torch.device("cuda", 0))
assert empty.numel() == 0 and empty.is_cuda
dist.barrier(); dist.destroy_process_group(); print("ok")
""")
    env = dict({k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")},
               MASTER_ADDR="127.0.0.1", MASTER_PORT="29672", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]
