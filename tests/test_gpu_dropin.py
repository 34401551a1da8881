"""GPU tests of the reference-shaped classes: they read like the reference's call sites
(spokestack/wakeword/tflite.py, utils/evaluate_models.py, utils/evaluate_tf_lite_opts.py) and
are checked against the CPU oracles."""
import os
import wave

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
TOL = 1e-4


@pytest.fixture(scope="module")
def oracle_dirs(assets):
    from oracle.tflite_interp import ModelDir
    return {m: ModelDir(os.path.join(assets, m)) for m in ("CRNN", "Wavenet", "CRNN_softmax")}


def test_tflite_model_three_files_like_the_reference(assets, oracle_dirs):
    """The exact call pattern of evaluate_models.py:76-86 / wakeword/tflite.py:183-231."""
    from spokestack.models.tensorflow import TFLiteModel
    rng = np.random.default_rng(1)
    for name in ("CRNN", "Wavenet", "CRNN_softmax"):
        d = os.path.join(assets, name)
        filt = TFLiteModel(model_path=os.path.join(d, "filter.tflite"))
        enc = TFLiteModel(model_path=os.path.join(d, "encode.tflite"))
        det = TFLiteModel(model_path=os.path.join(d, "detect.tflite"))
        assert (filt.input_details[0]["shape"][-1] - 1) * 2 == 512
        is_crnn = name.startswith("CRNN")
        T = enc.input_details[0]["shape"][2 if is_crnn else 1]
        assert T == (151 if is_crnn else 182)
        assert det.input_details[0]["shape"][-1] == (64 if is_crnn else 32)
        mag = np.abs(rng.normal(0, 1, (1, 257))).astype(np.float32)
        mel = filt(mag)[0]
        assert mel.shape == (1, 40)
        assert np.abs(mel - oracle_dirs[name].filter(mag)[0]).max() < 2e-5
        frames = rng.uniform(0, 6, (T, 40)).astype(np.float32)
        if is_crnn:
            x = np.expand_dims(np.expand_dims(np.array(frames).T, 0), -1)
        else:
            x = np.expand_dims(np.array(frames), 0)
        e = np.array(enc(x)).squeeze(0)
        want_e = oracle_dirs[name].encode(x)[0]
        assert e.shape == want_e.shape and np.abs(e - want_e).max() < TOL
        out = det(e)[0]
        want = oracle_dirs[name].detect(want_e)[0]
        assert out.shape == want.shape and np.abs(out - want).max() < TOL
        with pytest.raises(ValueError):
            enc(x[:, :, :-1] if is_crnn else x[:, :-1])
        with pytest.raises(ValueError):
            filt(mag.astype(np.float64))
    with pytest.raises(ValueError):
        TFLiteModel(model_path=os.path.join(assets, "CRNN", "missing.tflite"))


def test_filter_streams_like_the_reference(assets, oracle_dirs):
    """utils/tf_lite/filter.py:38-57 driven in 20 ms chunks, including carry across 'files'."""
    from tf_lite.filter import Filter
    from oracle import numpy_ref as NR
    rng = np.random.default_rng(2)
    m = oracle_dirs["CRNN"]
    for pre in (0.0, 0.97):
        f = Filter(pre_emphasis=pre, model_dir=os.path.join(assets, "CRNN"))
        ref = NR.RefFilter(lambda a: m.filter(a)[0], pre_emphasis=pre)
        assert f.num_outputs() == 40
        counts = []
        for n_chunks in (10, 10, 3):          # three "files": the ring is never reset (quirk C2)
            x = rng.normal(0, 0.1, n_chunks * 320).astype(np.float32)
            got, want = [], []
            for s in range(0, len(x), 320):
                a, b = x[s:s + 320].copy(), x[s:s + 320].copy()
                got += f.filter_frame(a)
                want += ref.filter_frame(b)
                np.testing.assert_array_equal(a, b)        # same in-place pre-emphasis
            counts.append(len(got))
            assert len(got) == len(want)
            assert np.abs(np.array(got) - np.array(want)).max() < TOL
        assert counts == [17, 20, 6]
    with pytest.raises(ValueError, match="Invalid fft_window_type"):
        Filter(fft_window_type="hamming", model_dir=os.path.join(assets, "CRNN"))


def _reference_trigger_posteriors(mdir, frames_i16, speech_flags, T, pidx):
    """Posterior sequence the reference WakewordTrigger computes (tflite.py:148-239)."""
    from oracle import numpy_ref as NR
    ring = NR.RefRing(512)
    hann = np.hanning(512)
    window = np.zeros((T, 40), np.float32)
    out = []
    for fr, sp in zip(frames_i16, speech_flags):
        x = NR.normalise_pcm(fr)
        for s in x:
            ring.write(s)
            if ring.is_full:
                if sp:
                    w = ring.read_all()
                    mag = np.abs(np.fft.rfft(w * hann, n=512)).astype(np.float32)
                    mel = mdir.filter(mag[None])[0][0]
                    window = np.concatenate([window[1:], mel[None]])
                    out.append(float(mdir.window(window)[pidx]))
                ring.rewind().seek(160)
    return out


@pytest.mark.parametrize("name,mtype", [("CRNN", "CRNN"), ("Wavenet", "Wavenet")])
def test_wakeword_trigger_in_pipeline(assets, oracle_dirs, name, mtype):
    from spokestack.pipeline import SpeechPipeline
    from spokestack.wakeword.tflite import WakewordTrigger
    rng = np.random.default_rng(5)
    n_ticks = 40
    frames = [np.clip(rng.normal(0, 3000, 320), -32768, 32767).astype(np.int16) for _ in range(n_ticks)]
    speech = [t >= 3 for t in range(n_ticks)]       # VAD rises at tick 3

    class Src:
        i = 0
        def start(self): pass
        def stop(self): pass
        def close(self): pass
        def read(self):
            f = frames[self.i]
            self.i += 1
            return f

    class Vad:
        def __call__(self, ctx, frame):
            ctx.is_speech = speech[src.i - 1]
        def close(self): pass

    seen = []
    wake = WakewordTrigger(model_dir=os.path.join(assets, name), model_type=mtype, posterior_threshold=2.0)
    orig = wake._sample

    def spy(ctx, frame):
        before = wake._posterior_max
        orig(ctx, frame)
        seen.append(wake._posterior_max)
    wake._sample = spy
    src = Src()
    pipe = SpeechPipeline(src, [Vad(), wake])
    pipe.start()
    for _ in range(n_ticks):
        pipe.step()
    T = 151 if name == "CRNN" else 182
    want = _reference_trigger_posteriors(oracle_dirs[name], frames, speech, T, 0 if name == "CRNN" else 1)
    assert len(want) > 30
    assert abs(max(want) - seen[-1]) < TOL          # running posterior_max (tflite.py:233-234)
    assert not pipe.context.is_active               # threshold 2.0 never fires
    with pytest.raises(ValueError):
        WakewordTrigger(model_dir=os.path.join(assets, name), model_type="LSTM")
    with pytest.raises(ValueError, match="Invalid fft_window_type"):
        WakewordTrigger(fft_window_type="hamming", model_dir=os.path.join(assets, name), model_type=mtype)


def test_wakeword_trigger_activates_and_resets(assets):
    from wwhip.context import SpeechContext
    from wwhip.wakeword import WakewordTrigger
    rng = np.random.default_rng(6)
    woke = []
    wake = WakewordTrigger(model_dir=os.path.join(assets, "CRNN"), model_type="crnn", posterior_threshold=-1.0,
                           on_wake=lambda: woke.append(1))
    ctx = SpeechContext()
    events = []
    ctx.add_handler("activate", lambda c: events.append("activate"))
    ctx.is_speech = True
    for _ in range(3):
        wake(ctx, np.clip(rng.normal(0, 3000, 320), -32768, 32767).astype(np.int16))
    assert ctx.is_active and events == ["activate"] and woke == [1]
    ctx.is_speech = False                      # VAD fall -> reset (tflite.py:143-146)
    wake(ctx, np.zeros(320, np.int16))
    assert wake._posterior_max == 0.0
    wake.close()


def _write_wav(path, pcm):
    with wave.open(path, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())


@pytest.mark.parametrize("name,mtype", [("CRNN_softmax", "CRNN"), ("Wavenet", "Wavenet")])
def test_get_posterior_matches_reference_flow(assets, oracle_dirs, tmp_path, name, mtype):
    from wwhip.evaluate import get_posterior
    from oracle import numpy_ref as NR
    rng = np.random.default_rng(8)
    files = []
    sigs = []
    for i, n in enumerate((30000, 41000)):
        pcm = np.clip(rng.normal(0, 2500, n), -32768, 32767).astype(np.int16)
        p = str(tmp_path / f"f{i}.wav")
        _write_wav(p, pcm)
        files.append(p)
        sigs.append(pcm.astype(np.float32) / np.float32(32768))
    mdir = oracle_dirs[name]
    T = 151 if mtype == "CRNN" else 182
    filt = NR.RefFilter(lambda a: mdir.filter(a)[0])            # ONE filter for all files (quirk C2)
    want_all = [NR.sliding_posteriors(filt, s.copy(), T, lambda w: float(mdir.window(w)[1])) for s in sigs]
    pos = get_posterior(os.path.join(assets, name), mtype, "false_negatives", files, 20, 16000)
    assert len(pos) == 2
    for got, want in zip(pos, want_all):
        assert abs(got - max(want)) < TOL
    neg = get_posterior(os.path.join(assets, name), mtype, "false_accepts", files, 20, 16000)
    flat = [v for w in want_all for v in w]
    assert len(neg) == len(flat)
    assert np.abs(np.array(neg) - np.array(flat)).max() < TOL


def test_far_frr_identical_to_reference_numbers(assets):
    from wwhip.evaluate import far_frr, frr_at_fa
    from oracle import numpy_ref as NR
    rng = np.random.default_rng(9)
    neg = np.clip(rng.beta(0.3, 3.0, 5000), 0, 1).astype(np.float32)
    neg[1000:1040] = 0.97
    neg[3000:3030] = 0.8
    pos = rng.beta(5, 1, 300).astype(np.float32)
    thr, frr, fa, cnt = far_frr(pos, neg, 300, 5000 * 0.02 / 3600, models_dir=os.path.join(assets, "CRNN"))
    wf, wa, wc, _ = NR.far_frr(pos, neg, 300, 5000 * 0.02 / 3600)
    np.testing.assert_array_equal(cnt, wc)           # FA counts identical
    np.testing.assert_allclose(fa, wa, rtol=1e-15)
    np.testing.assert_allclose(frr, wf, atol=1e-15)
    assert frr_at_fa(frr, fa, 0.5) == NR.frr_at_fa(wf, wa, 0.5) or (np.isnan(frr_at_fa(frr, fa, 0.5)) and np.isnan(NR.frr_at_fa(wf, wa, 0.5)))
    with pytest.raises(ValueError):
        far_frr(pos, neg[:10], 300, 1.0, models_dir=os.path.join(assets, "CRNN"))


def test_models_predict_one_window_per_clip(assets):
    from wwhip.evaluate import load_data, models_predict
    from wwhip.models import engine_for
    from oracle.cpu import CpuOracle
    rng = np.random.default_rng(10)
    feats = [rng.uniform(0, 6, (n, 40)).astype(np.float32) for n in (147, 200, 60)]
    eng = engine_for(os.path.join(assets, "CRNN_softmax"))
    X, y = load_data(feats, [1, 0, 0], eng.window, 40)
    preds, post = models_predict(eng, X)
    want = CpuOracle(eng.blob).forward(X)[:, 1]
    assert np.abs(post - want).max() < TOL
    assert preds == [1 if p >= 0.5 else 0 for p in want]


@pytest.mark.parametrize("name", ["CRNN_softmax", "Wavenet"])
def test_testset_evaluation_far_frr_identical(assets, name):
    """SURVEY 8(d) cfg-1 stand-in, small: FA counts / FRR identical to the CPU oracle flow."""
    from wwhip.evaluate import synth_testset, evaluate_testset
    from wwhip.models import engine_for
    from oracle.cpu import CpuOracle
    from oracle import numpy_ref as NR
    clips, labels = synth_testset(24, seed=3, min_s=0.9, max_s=2.2)
    labels[:5] = 1
    eng = engine_for(os.path.join(assets, name))
    res = evaluate_testset(eng, clips, labels)
    ora = CpuOracle(eng.blob)
    pidx = eng.posterior_index
    pos, neg = [], []
    for c, l in zip(clips, labels):
        mel = ora.logmel(np.concatenate((np.zeros(8000, np.int16), c, np.zeros(8000, np.int16))))
        p = ora.slide_forward(mel, 2)[:, pidx]
        (pos if l else neg).append(p)
    pos = np.array([p.max() for p in pos], np.float32)
    neg = np.concatenate(neg)
    assert np.abs(res["positives"] - pos).max() < TOL and np.abs(res["negatives"] - neg).max() < TOL
    # same posteriors in -> identical counts out (sweep itself is exact)
    wf, wa, wc, _ = NR.far_frr(res["positives"], res["negatives"], int(labels.sum()), res["hours"])
    np.testing.assert_array_equal(res["fa_count"], wc)
    np.testing.assert_allclose(res["frr"], wf, atol=1e-15)


def test_dataset_filter_writes_the_reference_feature_file(assets, oracle_dirs, tmp_path):
    """utils/filter_dataset_to_h5.py end to end: wavs + metadata JSON -> test.h5 -> load_h5.
    Expected features: the reference's per-chunk loop over ONE never-reset Filter (quirk C2)."""
    import json
    from wwhip.dataset import DatasetFilter
    from wwhip.evaluate import load_h5, open_h5
    from oracle import numpy_ref as NR
    rng = np.random.default_rng(17)
    data_dir = tmp_path / "audio"
    data_dir.mkdir()
    meta, sigs = [], []
    for i, n in enumerate((9000, 0, 700, 16321)):
        pcm = np.clip(rng.normal(0, 2500, n), -32768, 32767).astype(np.int16)
        _write_wav(str(data_dir / f"utt{i}.wav"), pcm)
        meta.append({"audio_file_path": f"utt{i}.wav", "is_hotword": int(i == 0), "worker_id": f"w{i % 2}"})
        sigs.append(pcm.astype(np.float32) / np.float32(32768))
    (tmp_path / "test.json").write_text(json.dumps(meta))

    def energy_vad(frame_bytes, sr):  # plug-in with webrtcvad's call signature
        return np.abs(np.frombuffer(frame_bytes, np.int16)).mean() > 1500

    df = DatasetFilter(str(tmp_path / "test.json"), os.path.join(assets, "CRNN"), str(data_dir), str(tmp_path / "out"),
                       vad=energy_vad)
    clips = df.filter_dataset_audio()
    assert df.dataset_file.endswith("test.h5") and os.path.isfile(df.dataset_file)

    # the reference loop (filter_dataset_to_h5.py:63-112), literally
    mdir = oracle_dirs["CRNN"]
    filt = NR.RefFilter(lambda a: mdir.filter(a)[0])
    want = {}
    for m, s in zip(meta, sigs):
        if len(s) == 0:
            continue
        feats = []
        for st in range(0, len(s), 320):
            fr = s[st:st + 320].copy()
            if len(fr) < 320:
                fr = np.pad(fr, (0, 320 - len(fr)), mode="constant")
            feats.extend(filt.filter_frame(fr))
        if feats:
            want[m["audio_file_path"].replace(".wav", "")] = np.array(feats)
    assert [c["file_name"] for c in clips] == list(want)
    with open_h5(df.dataset_file) as h5:
        assert list(h5.keys()) == sorted(want)
        for k, w in want.items():
            got = h5[k][()]
            assert got.shape == w.shape and got.dtype == np.float32
            assert np.abs(got - w).max() < TOL
        assert int(h5["utt0"].attrs["is_hotword"]) == 1 and int(h5["utt3"].attrs["is_hotword"]) == 0
        assert int(h5["utt0"].attrs["speaker"]) == 0 and int(h5["utt3"].attrs["speaker"]) == 1
        assert int(h5["utt0"].attrs["speech_start_ts"]) == 0 and int(h5["utt0"].attrs["speech_end_ts"]) == (8640 + 320) // 160  # the last, mostly padded chunk is below the VAD threshold
    X, y = load_h5(df.dataset_file, 151, 40)
    assert X.shape == (len(want), 151, 40) and y.tolist() == [1, 0, 0]


def test_superframe_smoothing_matches_the_reference_lattice(assets):
    """wwhip.wfst (ww_superframe_smooth) vs the pynini-free restatement of wwdetect/wfst.py."""
    from wwhip import wfst
    from oracle import numpy_ref as NR
    t1 = [[0.8, 0.2], [0.9, 0.1], [0.5, 0.5], [0.4, 0.6], [0.2, 0.8], [0.6, 0.4], [0.3, 0.7], [0.4, 0.6], [0.5, 0.5], [0.9, 0.1]]
    assert wfst.smooth(t1) == "other other other wakeword wakeword wakeword wakeword wakeword other other"
    rng = np.random.default_rng(5)
    p = rng.uniform(0.0, 1.0, (3000, 10)).astype(np.float32)
    p[::97, 3] = 1.0      # p_other == 0 -> infinite cost
    p[::89, 5] = 0.0
    pp = np.stack([np.abs(p - np.float32(1)), p], axis=2)  # as CRNN_files/tflite.py:249-251 builds it
    paths, wake = wfst.smooth_batch(pp)
    want = np.array([NR.wfst_smooth(x) for x in pp], np.uint8)
    np.testing.assert_array_equal(paths, want)
    np.testing.assert_array_equal(wake, want.any(axis=1))
    # logarithm on the device: same paths except where two path costs tie within an ulp
    paths_dev, _ = wfst.smooth_batch(pp, device_log=True)
    assert (paths_dev != want).any(axis=1).mean() < 2e-3
    for T in (1, 2, 33, 64):
        q = rng.uniform(0.01, 0.99, (50, T)).astype(np.float32)
        qq = np.stack([1 - q, q], axis=2)
        np.testing.assert_array_equal(wfst.smooth_batch(qq)[0], np.array([NR.wfst_smooth(x) for x in qq], np.uint8))
    with pytest.raises(ValueError):
        wfst.smooth_batch(np.zeros((1, 65, 2), np.float32))
    # detector wiring: ten posteriors per decision
    det = wfst.SuperframeDetector(10)
    fired = [det.push(x[1]) for x in t1]
    assert fired == [False] * 9 + [True]


def test_evaluator_command_line_flow(assets, oracle_dirs, tmp_path):
    """tools/evaluate_models.py = utils/evaluate_models.py main(): test.json -> concatenated negative wav ->
    cached posteriors -> FRR / FA-per-hour.  Its numbers are compared with the literal restatement of main() (one
    never-reset Filter per get_posterior call, the per-file loop, plot_FRR_FAR's sweep) on the op-by-op oracle; a second run
    must come from the pickle caches; a 2-rank run (torchrun, gloo, the ranks share the card: wake-word files dealt over the
    ranks, the long negative wav cut into two posterior ranges) must print the same arrays."""
    import json
    import shutil
    import subprocess
    import sys
    from oracle import numpy_ref as NR
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    data = tmp_path / "snips"
    (data / "audio_files").mkdir(parents=True)
    rng = np.random.default_rng(23)
    meta, pcms = [], []
    for i in range(6):
        pcm = np.clip(rng.normal(0, 2500, int(rng.integers(30000, 42000))), -32768, 32767).astype(np.int16)
        pcms.append(pcm)
        _write_wav(str(data / "audio_files" / f"c{i}.wav"), pcm)
        meta.append({"audio_file_path": f"audio_files/c{i}.wav", "is_hotword": int(i < 2), "worker_id": "w"})
    (data / "test.json").write_text(json.dumps(meta))
    models = tmp_path / "models"
    shutil.copytree(os.path.join(assets, "CRNN_softmax"), models)
    script = os.path.join(root, "tools", "evaluate_models.py")
    args = ["--model_type", "CRNN", "--models_dir", str(models) + "/", "--data_dir", str(data) + "/",
            "--eval_dir", str(tmp_path / "evaluation") + "/"]
    cmd = [sys.executable, script] + args
    out1 = json.loads(subprocess.run(cmd, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
    assert out1["num_wakewords"] == 2 and len(out1["FRR"]) == 100 and len(out1["FA_per_hour"]) == 100
    assert all(0.0 <= v <= 1.0 for v in out1["FRR"])
    far = tmp_path / "evaluation" / "not_hey_snips_long.wav"
    assert far.exists() and (models / "CRNN_all_wakeword.pkl").exists() and (models / "CRNN_no_wakeword.pkl").exists()
    # the negative wav holds the first num_wakewords (= 2) negative clips joined by 100 ms of silence
    from wwhip.evaluate import read_wav
    n_expect = len(pcms[2]) + len(pcms[3]) + 1600
    assert len(read_wav(str(far))) == n_expect
    assert abs(out1["fa_hours"] - n_expect / 16000 / 3600) < 1e-12
    # the restatement of main() on the oracle
    mdir = oracle_dirs["CRNN_softmax"]
    window_fn = lambda w: float(mdir.window(w)[1])  # noqa: E731
    filt = NR.RefFilter(lambda a: mdir.filter(a)[0])            # get_posterior(..., "false_negatives", wakeword_paths): ONE Filter
    want_pos = [max(NR.sliding_posteriors(filt, p.astype(np.float32) / np.float32(32768), 151, window_fn)) for p in pcms[:2]]
    filt = NR.RefFilter(lambda a: mdir.filter(a)[0])            # get_posterior(..., "false_accepts", [FAR_path]): a new one
    want_neg = NR.sliding_posteriors(filt, read_wav(str(far)), 151, window_fn)
    import pickle
    got_pos = np.atleast_1d(np.squeeze(np.array(pickle.load(open(models / "CRNN_all_wakeword.pkl", "rb")))))
    got_neg = np.array(pickle.load(open(models / "CRNN_no_wakeword.pkl", "rb")))
    assert len(got_neg) == len(want_neg) and np.abs(got_neg - np.array(want_neg)).max() < TOL
    assert np.abs(got_pos - np.array(want_pos)).max() < TOL
    wf, wa, wc, _ = NR.far_frr(np.array(want_pos, np.float32), np.array(want_neg, np.float32), 2, n_expect / 16000 / 3600)
    assert out1["FA_count"] == wc.tolist()
    np.testing.assert_array_equal(np.array(out1["FRR"]), wf)
    np.testing.assert_allclose(np.array(out1["FA_per_hour"]), wa, rtol=1e-15)
    out2 = json.loads(subprocess.run(cmd, capture_output=True, text=True, check=True).stdout.strip().splitlines()[-1])
    assert out2 == out1
    # two ranks, no caches
    for f in ("CRNN_all_wakeword.pkl", "CRNN_no_wakeword.pkl"):
        (models / f).unlink()
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29655", script] + args, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    out3 = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out3["world_size"] == 2
    assert {k: v for k, v in out3.items() if k != "world_size"} == {k: v for k, v in out1.items() if k != "world_size"}
    np.testing.assert_array_equal(np.array(pickle.load(open(models / "CRNN_no_wakeword.pkl", "rb"))), got_neg)


def test_fp16_weight_variant_and_opts_script(assets, tmp_path):
    """the float16-quantised model variant (constants rounded to fp16, fp32 arithmetic) and
    tools/evaluate_tf_lite_opts.py (= utils/evaluate_tf_lite_opts.py main) over an H5 test set."""
    import json
    import subprocess
    import sys
    from wwhip import h5min, weights
    from wwhip.engine import Engine
    from oracle.cpu import CpuOracle
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(31)
    for name in ("CRNN_softmax", "Wavenet"):
        e = Engine(os.path.join(assets, name), weights_fp16=True)
        ora = CpuOracle(weights.pack_blob(weights.quantize_fp16(weights.load_model_dir(os.path.join(assets, name)))))
        ref32 = CpuOracle(weights.pack_blob(weights.load_model_dir(os.path.join(assets, name))))
        wins = rng.uniform(0, 6.5, (12, e.window, 40)).astype(np.float32)
        got = e.forward(wins)
        assert np.abs(got - ora.forward(wins)).max() < TOL
        d = np.abs(got - ref32.forward(wins)).max()
        assert 1e-7 < d < 5e-2      # a different model (fp16 constants), but a close one
        e.close()
    # H5 test set -> the script
    clips = {f"clip{i:03d}": (rng.uniform(0, 6.5, (int(rng.integers(60, 220)), 40)).astype(np.float32),
                              {"is_hotword": int(i % 3 == 0), "speaker": i % 5, "speech_start_ts": -1, "speech_end_ts": -1})
             for i in range(40)}
    (tmp_path / "ds").mkdir()
    h5min.write_datasets(str(tmp_path / "ds" / "test.h5"), clips)
    import shutil
    models = tmp_path / "models"
    shutil.copytree(os.path.join(assets, "CRNN_softmax"), models)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "evaluate_tf_lite_opts.py"), "--tf_models_dir", str(models),
                        "--dataset_dir", str(tmp_path / "ds"), "--testset", "test.h5", "--model_type", "CRNN"],
                       capture_output=True, text=True, check=True)
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    for k in ("float32", "float16"):
        m = res[k]
        assert m["true_negative"] + m["false_positive"] + m["true_positive"] + m["false_negative"] == 40
    assert (models / "tf_lite_results.npy").exists()
    # expected float32 predictions straight from the engine
    from wwhip.evaluate import load_h5, models_predict
    X, y = load_h5(str(tmp_path / "ds" / "test.h5"), 151, 40)
    e = Engine(os.path.join(assets, "CRNN_softmax"))
    preds, _ = models_predict(e, X)
    e.close()
    assert res["float32"]["true_positive"] == int(((np.array(preds) == 1) & (y == 1)).sum())


def test_demo_pipeline_on_a_wav(assets, tmp_path):
    """tools/demo.py: the reference's demo.py stage list (VAD, wake word, activation timeout) over a wav."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rng = np.random.default_rng(2)
    pcm = np.clip(rng.normal(0, 3000, 48000), -32768, 32767).astype(np.int16)
    _write_wav(str(tmp_path / "in.wav"), pcm)
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "demo.py"), "--models_dir", os.path.join(assets, "Wavenet"),
                        "--model_type", "Wavenet", "--wav", str(tmp_path / "in.wav")], capture_output=True, text=True, check=True)
    assert "wake events at (s):" in r.stdout and "Script completed" in r.stdout


@pytest.mark.parametrize("model_type", ["CRNN", "Wavenet"])
def test_time_tf_models_script(model_type):
    """tools/time_tf_models.py: the reference's batch-1 timing harness (utils/time_tf_models.py) on the TFLiteModel surface -
    encode alone (what the reference's loop times, quirk C7), encode + detect, and the one-call form."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "time_tf_models.py"), "--model_type", model_type, "--num_runs", "5"],
                       capture_output=True, text=True, check=True)
    res = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert res["model_type"] == model_type and 0 < res["encode_only"] <= res["encode_and_detect"] * 1.5 and 0 < res["engine_forward"] < 0.01
    assert "Script completed" in r.stdout


def test_many_stream_pipeline_matches_single_stream_stages(assets):
    """SURVEY 8(f) rank 3 / BASELINE configs[4] as a PIPELINE: VadBank -> WakewordBank -> one ActivationTimeout per stream
    over 128 streams in lock step, against 128 independent single-stream chains
    VoiceActivityDetector -> WakewordTrigger -> ActivationTimeout driven the way SpeechPipeline._dispatch drives them
    (demo.py:29-36, spokestack/pipeline.py:25-28): same is_speech / is_active after every tick, same activate /
    deactivate events, same posteriors."""
    from wwhip.activation_timeout import ActivationTimeout
    from wwhip.context import SpeechContext
    from wwhip.vad import VadBank, VoiceActivityDetector
    from wwhip.wakeword import WakewordBank, WakewordTrigger
    S, TICKS, THR = 128, 90, 0.02
    mdir = os.path.join(assets, "CRNN")
    rng = np.random.default_rng(77)
    frames = np.clip(rng.normal(0, 2500, (TICKS, S, 320)), -32768, 32767).astype(np.int16)
    # scripted raw VAD decisions: runs of speech / non-speech of random length per stream
    raw = np.zeros((TICKS, S), bool)
    for s in range(S):
        t, v = 0, bool(rng.random() < 0.5)
        while t < TICKS:
            n = int(rng.integers(1, 25))
            raw[t:t + n, s] = v
            t, v = t + n, not v
    kw_vad = dict(frame_width=20, vad_rise_delay=40, vad_fall_delay=60)
    kw_to = dict(frame_width=20, min_active=60, max_active=200)

    def make_ctx(log):
        ctx = SpeechContext()
        for name in ("activate", "deactivate"):
            ctx.add_handler(name, (lambda n: (lambda c: log.append(n)))(name))
        return ctx

    # ---- the banked pipeline
    b_logs = [[] for _ in range(S)]
    b_ctx = [make_ctx(b_logs[s]) for s in range(S)]
    vad_bank = VadBank(S, **kw_vad)
    wake_bank = WakewordBank(S, mdir, posterior_threshold=THR)
    b_to = [ActivationTimeout(**kw_to) for _ in range(S)]
    b_state = np.zeros((TICKS, S, 2), bool)
    b_post = []
    for t in range(TICKS):
        speech = vad_bank.step(raw[t])
        for s in range(S):
            b_ctx[s].is_speech = bool(speech[s])
        post = wake_bank.step(b_ctx, frames[t])
        b_post.append(post.copy())
        for s in range(S):
            b_to[s](b_ctx[s], frames[t, s])
            b_state[t, s] = (b_ctx[s].is_speech, b_ctx[s].is_active)
    wake_bank.close()

    # ---- 128 single-stream chains, stage order of demo.py: vad, wake word, timeout
    n_active = 0
    for s in range(S):
        log = []
        ctx = make_ctx(log)
        tick = {"t": 0}
        vad = VoiceActivityDetector(classifier=lambda fb, sr: bool(raw[tick["t"], s]), **kw_vad)
        wake = WakewordTrigger(model_dir=mdir, model_type="CRNN", posterior_threshold=THR)
        timeout = ActivationTimeout(**kw_to)
        for t in range(TICKS):
            tick["t"] = t
            for stage in (vad, wake, timeout):
                stage(ctx, frames[t, s])
            assert (ctx.is_speech, ctx.is_active) == tuple(b_state[t, s]), (s, t)
        wake.close()
        assert log == b_logs[s], s
        n_active += "activate" in log
    # the script exercises both outcomes
    assert 4 <= n_active <= S - 4, n_active
    assert any("deactivate" in lg for lg in b_logs)
    assert np.isfinite(np.array(b_post)).all()


@pytest.mark.parametrize("fused", [True, False])
def test_banked_pipeline_on_a_context_bank_equals_the_per_context_form(assets, fused):
    """BASELINE configs[4] at the plugin surface, the fast form: SpeechPipelineBank([VadBank, WakewordBank, ActivationTimeoutBank])
    on a ContextBank - one library pass per stage and tick (fused=False), or ONE library call for the whole stage list
    (ww_pipeline_bank_step, the default for exactly this trio), events only for the streams that change - against the form the test
    above holds equal to 128 single-stream chains (a list of SpeechContext objects, one ActivationTimeout per stream): the same
    is_speech / is_active after every tick, the same activate / deactivate events per stream, the same posteriors, bit for bit."""
    from wwhip.activation_timeout import ActivationTimeout, ActivationTimeoutBank
    from wwhip.context import SpeechContext
    from wwhip.pipeline import SpeechPipelineBank
    from wwhip.vad import VadBank
    from wwhip.wakeword import WakewordBank
    S, TICKS, THR = 128, 90, 0.02
    mdir = os.path.join(assets, "CRNN")
    rng = np.random.default_rng(77)
    frames = np.clip(rng.normal(0, 2500, (TICKS, S, 320)), -32768, 32767).astype(np.int16)
    raw = np.zeros((TICKS, S), bool)
    for s in range(S):
        t, v = 0, bool(rng.random() < 0.5)
        while t < TICKS:
            n = int(rng.integers(1, 25))
            raw[t:t + n, s] = v
            t, v = t + n, not v
    kw_vad = dict(frame_width=20, vad_rise_delay=40, vad_fall_delay=60)
    kw_to = dict(frame_width=20, min_active=60, max_active=200)

    # ---- per-context form (as in test_many_stream_pipeline_matches_single_stream_stages)
    a_logs = [[] for _ in range(S)]
    a_ctx = []
    for s in range(S):
        c = SpeechContext()
        for name in ("activate", "deactivate"):
            c.add_handler(name, (lambda s, n: (lambda ctx: a_logs[s].append(n)))(s, name))
        a_ctx.append(c)
    vad_a, wake_a = VadBank(S, **kw_vad), WakewordBank(S, mdir, posterior_threshold=THR)
    to_a = [ActivationTimeout(**kw_to) for _ in range(S)]
    a_state = np.zeros((TICKS, S, 2), bool)
    a_post = []
    for t in range(TICKS):
        speech = vad_a.step(raw[t])
        for s in range(S):
            a_ctx[s].is_speech = bool(speech[s])
        a_post.append(wake_a.step(a_ctx, frames[t]))
        for s in range(S):
            to_a[s](a_ctx[s], frames[t, s])
            a_state[t, s] = (a_ctx[s].is_speech, a_ctx[s].is_active)
    wake_a.close()

    # ---- the pipeline on a ContextBank
    class Source:
        t = 0

        def read(self):
            self.t += 1
            return frames[self.t - 1]

        def start(self):
            pass

        def stop(self):
            pass

        def close(self):
            pass

    src = Source()
    woke = []
    pipe = SpeechPipelineBank(src, [VadBank(S, classifier=lambda f: raw[src.t - 1], **kw_vad),
                                    WakewordBank(S, mdir, posterior_threshold=THR, on_wake=lambda ids: woke.extend(int(i) for i in ids)),
                                    ActivationTimeoutBank(S, **kw_to)], S, fused=fused)
    assert (pipe._fused is not None) == fused
    b_logs = [[] for _ in range(S)]
    for s in range(0, S, 3):  # per-stream handlers on a third of the streams ...
        for name in ("activate", "deactivate"):
            pipe.context[s].add_handler(name, (lambda s, n: (lambda ctx: b_logs[s].append(n)))(s, name))
    all_b = [[] for _ in range(S)]
    pipe.event(lambda ctx: all_b[ctx._s].append("activate"), name="activate")      # ... and one handler for every stream
    pipe.event(lambda ctx: all_b[ctx._s].append("deactivate"), name="deactivate")
    wake_b = pipe._stages[1]
    pipe.start()
    for t in range(TICKS):
        pipe.step()
        assert np.array_equal(pipe.context.is_speech.astype(bool), a_state[t, :, 0]), t
        assert np.array_equal(pipe.context.is_active.astype(bool), a_state[t, :, 1]), t
        np.testing.assert_array_equal(wake_b._post, a_post[t])
    for s in range(S):
        assert all_b[s] == a_logs[s], s
        assert b_logs[s] == (a_logs[s] if s % 3 == 0 else [])
    assert sorted(woke) == sorted(s for s in range(S) for ev in a_logs[s] if ev == "activate")
    assert 4 <= sum("activate" in lg for lg in a_logs) <= S - 4 and any("deactivate" in lg for lg in a_logs)
    pipe.stop()
    pipe.cleanup()


def test_keyword_recognizer_matches_the_reference_loop(assets, oracle_dirs):
    """``KeywordRecognizer`` (spokestack/asr/keyword/tflite.py:99-184) - HIP front end, frames analysed
    only while the context is active, autoregressive encoder, detection on the falling edge of ``is_active`` - against the
    literal restatement of the reference class (oracle/numpy_ref.RefKeywordRecognizer, op-by-op filter graph).  The
    reference ships no keyword models: encoder and detector are deterministic stand-ins with TFLiteModel's protocol."""
    from spokestack.asr.keyword.tflite import KeywordRecognizer  # the reference's import path
    from wwhip.keyword import CallableModel
    from wwhip.context import SpeechContext
    from oracle import numpy_ref as NR
    rng = np.random.default_rng(55)
    MEL_LEN, ENC_LEN, ENC_W, CLASSES = 12, 9, 6, ["up", "down", "stop"]
    w_enc = rng.normal(0, 0.3, (40, ENC_W)).astype(np.float32)
    w_st = rng.normal(0, 0.3, (ENC_W, ENC_W)).astype(np.float32)
    w_det = rng.normal(0, 1.0, (ENC_W, len(CLASSES))).astype(np.float32)

    def encode(win, state):  # [1, MEL_LEN, 40], [1, ENC_W] -> ([1, ENC_W], [1, ENC_W])
        e = np.tanh(win[0].mean(axis=0) @ w_enc * 0.2 + state[0] @ w_st).astype(np.float32)[None]
        return e, (0.5 * state + 0.5 * e).astype(np.float32)

    def detect(win):  # [1, ENC_LEN, ENC_W] -> [[p_class...]]
        z = win[0].mean(axis=0) @ w_det
        p = np.exp(z - z.max())
        return [(p / p.sum()).astype(np.float32)[None]]

    enc_m = CallableModel(encode, [(1, MEL_LEN, 40), (1, ENC_W)], [(1, ENC_W), (1, ENC_W)])
    det_m = CallableModel(detect, [(1, ENC_LEN, ENC_W)], [(1, len(CLASSES))])
    with pytest.raises(ValueError):
        KeywordRecognizer(["a", "b"], encode_model=enc_m, detect_model=det_m, filter_model_dir=os.path.join(assets, "CRNN"))
    with pytest.raises(ValueError):
        KeywordRecognizer(CLASSES, fft_window_type="hamming", encode_model=enc_m, detect_model=det_m,
                          filter_model_dir=os.path.join(assets, "CRNN"))

    def run(stage, thr_note):
        ctx = SpeechContext()
        events = []
        for name in ("recognize", "timeout"):
            ctx.add_handler(name, (lambda n: (lambda c: events.append((n, c.transcript, float(c.confidence)))))(name))
        trace = []
        for t in range(len(frames)):
            ctx.is_active = bool(active[t])
            stage(ctx, frames[t])
            trace.append(len(events))
        return events, trace

    frames = np.clip(rng.normal(0, 4000, (160, 320)), -32768, 32767).astype(np.int16)
    active = np.zeros(160, bool)
    for a, b in ((5, 40), (41, 43), (60, 61), (90, 150)):   # long, very short (no frame completes), one tick, long
        active[a:b] = True
    mdir = oracle_dirs["CRNN"]
    seen = set()
    for thr in (0.2, 0.995):
        got = run(KeywordRecognizer(CLASSES, model_dir=os.path.join(assets, "CRNN"), posterior_threshold=thr,
                                    encode_model=enc_m, detect_model=det_m), thr)
        ref = NR.RefKeywordRecognizer(CLASSES, lambda a: mdir.filter(a)[0], encode, lambda w: detect(w), MEL_LEN, 40, (1, ENC_W),
                                      ENC_LEN, ENC_W, posterior_threshold=thr)
        want = run(ref, thr)
        assert got[1] == want[1]                                   # the same tick fires each event
        assert len(got[0]) == len(want[0]) == 4
        for (gn, gt, gc), (wn, wt, wc) in zip(got[0], want[0]):
            assert (gn, gt) == (wn, wt) and abs(gc - wc) < TOL
        seen |= {e[0] for e in got[0]}
    assert seen == {"recognize", "timeout"}


def test_entry_points_leave_the_callers_device_alone(assets):
    """Every C entry point switches to its context's device only for the duration of the call (ww_device_scope): with
    torch on ANOTHER current device the engine on device 0 still computes correctly and torch's current device is
    unchanged.  Needs two GPUs (skipped on the one-GPU boxes); with one GPU the same calls at least must not move it."""
    import torch
    from wwhip import _lib
    from wwhip.engine import Engine
    from wwhip.evaluate import clip_posteriors, synth_testset
    from oracle.cpu import CpuOracle
    n_dev = torch.cuda.device_count()
    other = 1 if n_dev >= 2 else 0
    torch.cuda.set_device(other)
    try:
        ctx = _lib.Context(0)
        eng = Engine(os.path.join(assets, "CRNN"), device=0, ctx=ctx)
        assert torch.cuda.current_device() == other
        rng = np.random.default_rng(2)
        wins = rng.uniform(0, 6, (5, eng.window, 40)).astype(np.float32)
        got = eng.forward(wins)
        mel = eng.logmel([np.clip(rng.normal(0, 3000, 9000), -32768, 32767).astype(np.int16)])[0]
        clips, _ = synth_testset(6, seed=9)
        p_one, _ = clip_posteriors(eng, clips)          # torch buffers must land on the ENGINE's device
        assert torch.cuda.current_device() == other
        ora = CpuOracle(eng.blob)
        assert np.abs(got - ora.forward(wins)).max() < TOL and mel.shape[1] == 40 and len(p_one) == 6
        eng.close()
        ctx.close()
        assert torch.cuda.current_device() == other
    finally:
        torch.cuda.set_device(0)
    if n_dev < 2:
        pytest.skip("one GPU: only the no-op half of the check could run")


def test_get_posterior_edge_cases(assets, tmp_path):
    """get_posterior off the beaten path, against the C oracle over the same padded stream(s): a custom loader that returns
    float32 samples (the front end's float entry point), `carry_over=False` (a fresh ring per file), a 25 ms `frame_width`
    (400-sample chunks: not a multiple of the hop), a file too short for a window - skipped in the negative stream, `ValueError`
    for a positive clip (the reference's `np.max` of an empty list) - and an empty file list."""
    from wwhip.evaluate import get_posterior, read_wav, StreamPlan
    from wwhip.models import engine_for
    from oracle.cpu import CpuOracle
    mdir = os.path.join(assets, "CRNN_softmax")
    eng = engine_for(mdir)
    ora = CpuOracle(eng.blob)
    rng = np.random.default_rng(77)
    pcms = [np.clip(rng.normal(0, 2500, n), -32768, 32767).astype(np.int16) for n in (30000, 3000, 41000)]
    files = []
    for i, p in enumerate(pcms):
        files.append(str(tmp_path / f"e{i}.wav"))
        _write_wav(files[-1], p)

    def oracle_stream(lens_pcms, fl, carry):
        plan = StreamPlan([len(p) for p in lens_pcms], eng.window, fl, carry_over=carry)
        out = []
        if carry:
            whole = np.zeros(int(plan.padded.sum()), np.int16)
            for k, p in enumerate(lens_pcms):
                whole[plan.pos[k] + 8000: plan.pos[k] + 8000 + len(p)] = p
            mel = ora.logmel(whole, 32768.0, False)
            for k in range(len(lens_pcms)):
                rows = mel[plan.F[k]: plan.F[k] + plan.n_frames[k]]
                out.append(ora.slide_forward(rows, 2)[:plan.n_win[k], 1] if plan.n_win[k] else np.zeros(0, np.float32))
        else:
            for k, p in enumerate(lens_pcms):
                one = np.zeros(int(plan.padded[k]), np.int16)
                one[8000:8000 + len(p)] = p
                out.append(ora.slide_forward(ora.logmel(one, 32768.0, False), 2)[:plan.n_win[k], 1])
        return plan, out

    for fw, carry, loader in ((20, True, None), (20, False, None), (25, True, None), (20, True, lambda p: read_wav(p))):
        plan, want = oracle_stream(pcms, 16 * fw, carry)
        assert plan.n_win[1] == 0 and plan.n_win[0] > 0
        neg = np.array(get_posterior(mdir, "CRNN", "false_accepts", files, fw, 16000, loader=loader, carry_over=carry), np.float32)
        flat = np.concatenate(want)
        assert len(neg) == len(flat) == plan.total
        assert np.abs(neg - flat).max() < TOL, (fw, carry, float(np.abs(neg - flat).max()))
    # a STEREO wav among mono ones (librosa averages the channels: float32 samples): every file is then brought to librosa's
    # float scale - the mono files' int16 / 32768 - instead of mixing raw PCM with floats in one launch
    st = str(tmp_path / "stereo.wav")
    with wave.open(st, "wb") as w:
        w.setnchannels(2); w.setsampwidth(2); w.setframerate(16000)
        w.writeframes(np.stack([pcms[2], pcms[2]], axis=1).tobytes())      # both channels = clip 2: their mean is clip 2
    neg_mixed = np.array(get_posterior(mdir, "CRNN", "false_accepts", [files[0], files[1], st], 20, 16000), np.float32)
    neg_mono = np.array(get_posterior(mdir, "CRNN", "false_accepts", files, 20, 16000), np.float32)
    assert len(neg_mixed) == len(neg_mono) and np.abs(neg_mixed - neg_mono).max() < 1e-5
    pos = get_posterior(mdir, "CRNN", "false_negatives", [files[0], files[2]], 20, 16000)
    plan, want = oracle_stream([pcms[0], pcms[2]], 320, True)
    assert np.abs(np.array(pos) - np.array([w.max() for w in want])).max() < TOL
    with pytest.raises(ValueError):
        get_posterior(mdir, "CRNN", "false_negatives", files, 20, 16000)     # the 3,000-sample clip yields no window
    assert get_posterior(mdir, "CRNN", "false_accepts", [], 20, 16000) == []
    with pytest.raises(ValueError):
        get_posterior(mdir, "LSTM", "false_accepts", files, 20, 16000)
