"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the
committed golden fixtures.  Tolerances follow BASELINE.json: posteriors within 1e-4 (fp32);
integer results (frame counts, FA counts) exact."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# every model artefact the reference ships (SURVEY Appendix D file inventory): tf_lite_models/{CRNN,Wavenet},
# utils/Wavenet_files (Wavenet_alt), wwdetect/CRNN/models/Arik_CRNN_data_{original,nosilence,nosilence_enhanced}
# (CRNN_softmax, CRNN_nosilence, CRNN_nosilence_enhanced) and utils/CRNN_files/*_old.tflite (CRNN_old: other conv geometry)
MODELS = ["CRNN", "CRNN_softmax", "Wavenet", "Wavenet_alt", "CRNN_nosilence", "CRNN_nosilence_enhanced", "CRNN_old"]
TOL_POST = 1e-4   # north_star: per-frame posteriors within 1e-4 fp32
TOL_MEL = 1e-4    # log-mel (SURVEY 7 minimum slice)


@pytest.fixture(scope="module")
def engines(assets):
    from wwhip.engine import Engine
    out = {m: Engine(os.path.join(assets, m)) for m in MODELS}
    yield out
    for e in out.values():
        e.close()


@pytest.fixture(scope="module")
def oracles(engines):
    from oracle.cpu import CpuOracle
    return {m: CpuOracle(e.blob) for m, e in engines.items()}


def test_native_library_is_the_path(engines):
    from wwhip import _lib
    assert os.path.isfile(_lib.LIB_PATH)
    with open("/proc/self/maps") as f:
        assert "libwwhip.so" in f.read()


# ---------------------------------------------------------------- front end
def test_stft_magnitude_vs_numpy(engines):
    rng = np.random.default_rng(3)
    frames = rng.normal(0, 0.2, (37, 512)).astype(np.float32)
    frames[0] = 0.0
    frames[1] = 0.5 * np.sin(2 * np.pi * 1000 * np.arange(512) / 16000)
    want = np.abs(np.fft.rfft(frames.astype(np.float64) * np.hanning(512), n=512)).astype(np.float32)
    got = engines["CRNN"].stft_mag(frames, precise=True)
    np.testing.assert_allclose(got, want, rtol=2e-6, atol=1e-9)
    fast = engines["CRNN"].stft_mag(frames, precise=False)
    assert np.abs(fast - want).max() < 2e-4 * max(1.0, float(want.max()))


@pytest.mark.parametrize("div,clip,pre", [(32767.0, True, 0.0), (32768.0, False, 0.97)])
def test_logmel_golden(engines, golden, div, clip, pre):
    from wwhip.engine import frontend_params
    z = np.load(os.path.join(golden, "frontend.npz"))
    names = ["noise_chirp", "quiet", "silence", "fullscale", "ragged"]
    pcm = [z[n + ".pcm"] for n in names]
    mels = engines["CRNN"].logmel(pcm, frontend_params(div, clip, pre, 160, True))
    for n, got in zip(names, mels):
        want = z[f"{n}.div{int(div)}.mel"]
        assert got.shape == want.shape, n
        assert np.abs(got - want).max() < TOL_MEL, (n, np.abs(got - want).max())


def test_logmel_vs_oracle_ragged_batch(engines, oracles):
    rng = np.random.default_rng(11)
    lens = [24000, 511, 512, 513, 100, 0, 671, 672, 40000, 8000]
    pcm = [np.clip(rng.normal(0, 3000, n), -32768, 32767).astype(np.int16) for n in lens]
    mels = engines["Wavenet"].logmel(pcm)
    for n, p, got in zip(lens, pcm, mels):
        want = oracles["Wavenet"].logmel(p)
        assert got.shape == want.shape == (max(0, (n - 512) // 160 + 1) if n >= 512 else 0, 40)
        if len(want):
            assert np.abs(got - want).max() < TOL_MEL


def test_logmel_float_input_matches_int16(engines):
    from wwhip.engine import frontend_params
    rng = np.random.default_rng(12)
    pcm = np.clip(rng.normal(0, 3000, 9000), -32768, 32767).astype(np.int16)
    a = engines["CRNN"].logmel([pcm], frontend_params(32768.0, False, 0.5))[0]
    b = engines["CRNN"].logmel([pcm.astype(np.float32) / np.float32(32768.0)], frontend_params(32768.0, False, 0.5))[0]
    np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("pre,hop,as_float", [(0.0, 160, False), (0.97, 160, False), (0.0, 100, True), (0.5, 512, False), (0.0, 7, False)])
def test_logmel_rows_across_clip_boundaries(engines, oracles, pre, hop, as_float):
    """logmel_rows_kernel numbers mel rows through the whole launch and gives a wave four consecutive rows, whatever clips
    they belong to: a batch must give, bit for bit, what its clips give one at a time (where every wave holds rows of one
    clip), including clips of 1-3 frames (several boundaries inside one wave), frame-less clips in between, odd sample
    offsets (the unaligned staging) and the ragged end of the buffer."""
    from wwhip.engine import frontend_params
    rng = np.random.default_rng(77)
    W = 512
    lens = [W + 146 * hop, W, W + hop, 0, W + 2 * hop + 1, 37, W + 5 * hop + 3, W + 3 * hop, W - 1, W + 11 * hop + hop // 2, W + 4 * hop]
    pcm = [np.clip(rng.normal(0, 4000, n), -32768, 32767).astype(np.int16) for n in lens]
    if as_float:
        pcm = [p.astype(np.float32) / np.float32(32768.0) for p in pcm]
    fp = frontend_params(32768.0, False, pre, hop, True)
    eng = engines["CRNN"]
    batch = eng.logmel(pcm, fp)
    for n, p, got in zip(lens, pcm, batch):
        nf = (n - W) // hop + 1 if n >= W else 0
        assert got.shape == (nf, 40)
        alone = eng.logmel([p], fp)[0]
        np.testing.assert_array_equal(got, alone)
    if not as_float and hop == 160:
        for p, got in zip(pcm, batch):
            if len(got):
                assert np.abs(got - oracles["CRNN"].logmel(p, 32768.0, False, pre)).max() < TOL_MEL


def test_logmel_equal_clips_arithmetic_lookup_equals_offset_tables(engines):
    """ww_clips_forward_dev tells the front end that its clips are equal and back to back (row -> clip by arithmetic);
    ww_logmel on the same clips walks the offset tables: the same posteriors, bit for bit, for a clip length whose frame count
    is not a multiple of four (every clip boundary falls inside a wave)."""
    import torch
    from wwhip.engine import frontend_params
    rng = np.random.default_rng(78)
    eng = engines["CRNN"]
    for samples in (24000, 512 + 160 * 6, 700):
        pcm = np.clip(rng.normal(0, 3000, (9, samples)), -32768, 32767).astype(np.int16)
        fp = frontend_params(32767.0, True, 0.0, 160, True)
        d = torch.from_numpy(pcm).cuda()
        out = torch.empty((9, eng.n_out), dtype=torch.float32, device="cuda")
        eng.clips_forward_dev(d.data_ptr(), 9, samples, out.data_ptr(), fp)
        eng.ctx.synchronize()
        mels = eng.logmel(list(pcm), fp)
        wins = np.zeros((9, eng.window, 40), np.float32)
        for i, m in enumerate(mels):
            wins[i, :min(len(m), eng.window)] = m[:eng.window]
        np.testing.assert_array_equal(out.cpu().numpy(), eng.forward(wins))


# ---------------------------------------------------------------- models
@pytest.mark.parametrize("name", MODELS)
def test_forward_golden(engines, golden, name):
    z = np.load(os.path.join(golden, "models.npz"))
    out, enc = engines[name].forward(z[name + ".windows"], want_enc=True)
    assert np.abs(out - z[name + ".det64"]).max() < TOL_POST
    assert np.abs(out - z[name + ".det32"]).max() < TOL_POST
    want_enc = z[name + ".enc32"]
    assert np.abs(enc.reshape(want_enc.shape) - want_enc).max() < 1e-4


@pytest.mark.parametrize("name", MODELS)
def test_forward_vs_oracle_batch(engines, oracles, name):
    rng = np.random.default_rng(21)
    e = engines[name]
    wins = rng.uniform(0, 6.5, (70, e.window, 40)).astype(np.float32)
    wins[3] = 0
    wins[4, 100:] = 0
    got = e.forward(wins)
    want = oracles[name].forward(wins)
    assert got.shape == want.shape
    assert np.abs(got - want).max() < TOL_POST


@pytest.mark.parametrize("name", ["CRNN", "Wavenet", "CRNN_old"])
def test_slide_forward_vs_oracle(engines, oracles, name):
    rng = np.random.default_rng(22)
    e = engines[name]
    mel = rng.uniform(0, 6.5, (e.window + 61, 40)).astype(np.float32)
    got = e.slide_forward(mel, hop=2)
    want = oracles[name].slide_forward(mel, 2)
    assert got.shape == want.shape == (31, e.n_out)
    assert np.abs(got - want).max() < TOL_POST
    assert e.slide_forward(mel[: e.window - 1], hop=2).shape == (0, e.n_out)


@pytest.mark.parametrize("sub,name", [("", "CRNN_softmax"), ("nosilence", "CRNN_nosilence"),
                                      ("nosilence_enhanced", "CRNN_nosilence_enhanced")])
def test_forward_vs_keras_checkpoint(engines, golden, sub, name):
    """The HIP path against the Keras checkpoints the three Arik_CRNN_data_* .tflite pairs were converted from,
    evaluated with Keras-documented layer semantics in float64 (oracle/keras_ref.py)."""
    from oracle import keras_ref
    rng = np.random.default_rng(23)
    wins = rng.uniform(0, 6.5, (9, 151, 40)).astype(np.float32)
    wins[0] = 0
    wins[1, 60:] = 0
    kd = os.path.join(golden, "keras_h5", sub)
    want, want_enc = keras_ref.crnn_forward(os.path.join(kd, "encode.h5"), os.path.join(kd, "detect.h5"), wins)
    got, enc = engines[name].forward(wins, want_enc=True)
    assert np.abs(got - want).max() < TOL_POST
    assert np.abs(enc.reshape(want_enc.shape) - want_enc).max() < 1e-4


@pytest.mark.parametrize("name", ["Wavenet", "Wavenet_alt"])
def test_wavenet_split_bf16_mode(assets, oracles, golden, name):
    """ww_model_set_precision(BF16X3): the 24 gated blocks on bf16 MFMA with split operands stay
    within the north-star tolerance of the fp32 oracle (error model: tools/bf16x3_error.py)."""
    from wwhip.engine import Engine
    e = Engine(os.path.join(assets, name), precision="bf16x3")
    try:
        rng = np.random.default_rng(29)
        wins = rng.uniform(0, 6.5, (37, e.window, 40)).astype(np.float32)
        wins[3] = 0
        wins[4, 100:] = 0
        wins[5] = np.clip(rng.normal(3, 1.5, (e.window, 40)), 0, 8)
        got, enc = e.forward(wins, want_enc=True)
        want, want_enc = oracles[name].forward(wins, want_enc=True)
        assert np.abs(got - want).max() < 2e-5          # measured 4e-6; north star 1e-4
        assert np.abs(enc - want_enc).max() < 1e-3 * max(1.0, float(np.abs(want_enc).max()))
        z = np.load(os.path.join(golden, "models.npz"))
        out = e.forward(z[name + ".windows"])
        assert np.abs(out - z[name + ".det64"]).max() < 2e-5
        # sliding mode goes through the same kernel
        mel = rng.uniform(0, 6.5, (e.window + 40, 40)).astype(np.float32)
        assert np.abs(e.slide_forward(mel, 2) - oracles[name].slide_forward(mel, 2)).max() < 2e-5
        e.set_precision("fp32")
        assert np.abs(e.forward(wins) - want).max() < 2e-6
        # a chip-filling batch, twice: every workgroup against the fp32 kernel (catches ordering hazards
        # between waves that a handful of windows never exercises)
        big = rng.uniform(0, 6.5, (700, e.window, 40)).astype(np.float32)
        big[::7, 140:] = 0
        ref = e.forward(big)
        e.set_precision("bf16x3")
        for _ in range(2):
            assert np.abs(e.forward(big) - ref).max() < 2e-5
        with pytest.raises(ValueError):
            e.set_precision("bf16")
    finally:
        e.close()


@pytest.mark.parametrize("name", ["CRNN", "CRNN_softmax"])
def test_crnn_large_batch_path(engines, oracles, name):
    """Above 1,024 windows per launch the CRNN runs as crnn_fused_kernel<front> + a tail kernel for the recurrences:
    gru_tail16_kernel (option crnn_tail_mfma = 2 here; by default from 9,216 windows per launch on: sixteen windows per workgroup,
    recurrent products and the layer-2 projection on v_mfma_f32_16x16x4_f32, a partial last workgroup: 1,500 = 93 x 16 + 12)
    and gru_tail_kernel (one window per workgroup on the vector ALU; options 0 and, at this size, 1).  Every form associates
    its sums the same way (round 4), so all of them equal the one-kernel path BIT FOR BIT - a posterior does not depend on how
    its window was dispatched; the oracle's within tolerance - windows with partial validity included, encoder output too."""
    rng = np.random.default_rng(31)
    e = engines[name]
    wins = rng.uniform(0, 6.5, (1500, e.window, 40)).astype(np.float32)
    wins[::13, 120:] = 0
    wins[7] = 0
    small = np.concatenate([e.forward(wins[i:i + 500]) for i in range(0, 1500, 500)])  # three fused launches
    idx = rng.choice(1500, 96, replace=False)
    want, want_enc = oracles[name].forward(wins[idx], want_enc=True)
    try:
        for mfma in (2, 1, 0):
            e.set_option("crnn_tail_mfma", mfma)
            big, big_enc = e.forward(wins, want_enc=True)       # one launch of 1,500 windows: front + tail
            np.testing.assert_array_equal(big, small)
            assert np.abs(big[idx] - want).max() < TOL_POST
            assert np.abs(big_enc[idx].reshape(want_enc.shape) - want_enc).max() < 1e-4
    finally:
        e.set_option("crnn_tail_mfma", 1)
    with pytest.raises(ValueError):
        e.set_option("no_such_option", 1)


def test_retired_precision_mode_is_refused(engines):
    """Round 1's experimental bf16x6 projection mode (WW_PRECISION value 2) is gone: asking for it is an error."""
    from wwhip import _lib
    e = engines["CRNN"]
    assert _lib.load().ww_model_set_precision(e.handle, 2) == _lib.WW_EINVAL
    with pytest.raises(ValueError):
        e.set_precision("bf16x6")


@pytest.mark.parametrize("name", ["CRNN", "CRNN_softmax", "CRNN_nosilence_enhanced"])
def test_crnn_split_bf16_mode(assets, oracles, golden, name):
    """ww_model_set_precision(BF16X3) on a CRNN: conv and layer-1 projection on bf16 MFMA with split operands
    (crnn_fused_bf16_kernel) stay within the north-star tolerance of the fp32 oracle, in both the one-kernel form and the
    front + tail form of large batches; fp32 remains the default and is restored exactly."""
    from wwhip.engine import Engine
    e = Engine(os.path.join(assets, name), precision="bf16x3")
    try:
        rng = np.random.default_rng(41)
        wins = rng.uniform(0, 6.5, (300, e.window, 40)).astype(np.float32)
        wins[3] = 0
        wins[4, 100:] = 0
        wins[5] = np.clip(rng.normal(3, 1.5, (e.window, 40)), 0, 8)
        got, enc = e.forward(wins, want_enc=True)
        want, want_enc = oracles[name].forward(wins, want_enc=True)
        assert np.abs(got - want).max() < 2e-5          # north star 1e-4
        assert np.abs(enc.reshape(want_enc.shape) - want_enc).max() < 2e-4
        z = np.load(os.path.join(golden, "models.npz"))
        assert np.abs(e.forward(z[name + ".windows"]) - z[name + ".det64"]).max() < 2e-5
        mel = rng.uniform(0, 6.5, (e.window + 40, 40)).astype(np.float32)
        assert np.abs(e.slide_forward(mel, 2) - oracles[name].slide_forward(mel, 2)).max() < 2e-5
        big = rng.uniform(0, 6.5, (1300, e.window, 40)).astype(np.float32)   # > 1,024 windows: front + tail kernels
        big[::7, 140:] = 0
        got_big = e.forward(big)
        np.testing.assert_array_equal(got_big[:300], e.forward(big[:300]))   # front + gru_tail_kernel: the fused kernel's arithmetic
        with e.options(crnn_tail_mfma=2):  # gru_tail16_kernel sums the recurrent products in another order
            assert np.abs(e.forward(big)[:300] - got_big[:300]).max() < 2e-6
        # ragged clips: windows with valid < T rows (zero padded in the kernel's staging) and sliding windows
        from wwhip.evaluate import clip_posteriors, synth_testset
        clips, _ = synth_testset(40, seed=5, min_s=0.6, max_s=2.4)
        one_b, slide_b = clip_posteriors(e, clips)
        e.set_precision("fp32")
        one_f, slide_f = clip_posteriors(e, clips)
        assert np.abs(one_b - one_f).max() < 5e-5 and np.abs(np.concatenate(slide_b) - np.concatenate(slide_f)).max() < 5e-5
        ref = e.forward(big)
        assert np.abs(got_big - ref).max() < 5e-5      # measured 2.5e-5 over 1,300 windows
        assert np.abs(e.forward(wins) - want).max() < 2e-6
    finally:
        e.close()


def test_forward_rejects_bad_shape(engines):
    with pytest.raises(ValueError):
        engines["CRNN"].forward(np.zeros((2, 150, 40), np.float32))


# ---------------------------------------------------------------- evaluator
@pytest.mark.parametrize("case", ["short", "long", "exact30"])
def test_far_frr_golden(engines, golden, case):
    z = np.load(os.path.join(golden, "evaluator.npz"))
    thr = np.arange(0.5, 0.99999, 0.005)
    frr, fa, cnt, sm = engines["CRNN"].far_frr(z[case + ".pos"], z[case + ".neg"], thr, 200, float(z[case + ".hours"][0]),
                                               want_smoothed=True)
    np.testing.assert_allclose(sm, z[case + ".smoothed"], rtol=0, atol=1e-15)
    np.testing.assert_array_equal(cnt, z[case + ".cnt"])
    np.testing.assert_allclose(fa, z[case + ".far"], rtol=1e-15)
    np.testing.assert_allclose(frr, z[case + ".frr"], rtol=0, atol=1e-15)


# ---------------------------------------------------------------- whole path, device resident
@pytest.mark.parametrize("name", ["CRNN", "Wavenet"])
def test_clips_forward_dev(engines, oracles, name):
    import torch
    from wwhip.engine import frontend_params
    rng = np.random.default_rng(31)
    e = engines[name]
    B, S = 33, 24000
    pcm = np.clip(rng.normal(0, 2500, (B, S)), -32768, 32767).astype(np.int16)
    d_pcm = torch.from_numpy(pcm).cuda()
    d_out = torch.zeros((B, e.n_out), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for _ in range(3):
        e.clips_forward_dev(d_pcm.data_ptr(), B, S, d_out.data_ptr(), frontend_params())
    e.ctx.synchronize()
    got = d_out.cpu().numpy()
    want = []
    for c in pcm:
        mel = oracles[name].logmel(c)
        win = np.zeros((e.window, 40), np.float32)
        win[: min(len(mel), e.window)] = mel[: e.window]
        want.append(oracles[name].forward(win)[0])
    assert np.abs(got - np.array(want)).max() < TOL_POST


# ---------------------------------------------------------------- streaming
@pytest.mark.parametrize("name", ["CRNN", "Wavenet"])
def test_stream_bank_matches_batch_path(engines, oracles, name):
    from wwhip.engine import StreamBank
    rng = np.random.default_rng(41)
    e = engines[name]
    S, ticks = 5, 90
    pcm = np.clip(rng.normal(0, 2500, (S, ticks * 320)), -32768, 32767).astype(np.int16)
    bank = StreamBank(e, S)
    posts = [[] for _ in range(S)]
    speech = np.ones(S, np.uint8)
    for t in range(ticks):
        p, n = bank.step(pcm[:, t * 320:(t + 1) * 320], speech)
        for s in range(S):
            posts[s] += [float(p[s, k]) for k in range(n[s])]
    bank.close()
    pidx = e.posterior_index
    for s in range(S):
        mel = oracles[name].logmel(pcm[s])
        # streaming semantics (tflite.py:96-97,187-188): window starts as zeros, slides by 1
        hist = np.concatenate([np.zeros((e.window, 40), np.float32), mel])
        want = oracles[name].slide_forward(hist, 1)[1:, pidx]
        assert len(posts[s]) == len(mel) == len(want)
        assert np.abs(np.array(posts[s]) - want).max() < TOL_POST


def test_fp32_frontend_mode_meets_posterior_tolerance(engines, oracles, golden):
    """precise=0 (fp32 butterflies): log-mel within 2e-4 even on full-scale tones, posteriors
    within the 1e-4 north-star tolerance of the fp64-FFT oracle."""
    from wwhip.engine import frontend_params
    z = np.load(os.path.join(golden, "frontend.npz"))
    for n in ["noise_chirp", "quiet", "silence", "fullscale", "ragged"]:
        got = engines["CRNN"].logmel([z[n + ".pcm"]], frontend_params(precise=False))[0]
        assert np.abs(got - z[n + ".div32767.mel"]).max() < 2e-4, n
    rng = np.random.default_rng(77)
    t = np.arange(24000) / 16000.0
    sig = rng.normal(0, 30, (8, 24000)) + 20000 * np.sin(2 * np.pi * 1000 * t)
    pcm = np.clip(np.rint(sig), -32768, 32767).astype(np.int16)
    for name in ("CRNN", "Wavenet"):
        e = engines[name]
        mels = e.logmel(list(pcm), frontend_params(precise=False))
        wins = np.zeros((len(pcm), e.window, 40), np.float32)
        ref = np.zeros_like(wins)
        for i, (m, p) in enumerate(zip(mels, pcm)):
            r = oracles[name].logmel(p)
            wins[i, :min(len(m), e.window)] = m[:e.window]
            ref[i, :min(len(r), e.window)] = r[:e.window]
        assert np.abs(e.forward(wins) - oracles[name].forward(ref)).max() < TOL_POST


# ---------------------------------------------------------------- BASELINE sizes: size-independent properties
@pytest.mark.parametrize("name,precision", [("CRNN", "fp32"), ("Wavenet", "fp32"), ("Wavenet", "bf16x3")])
def test_full_config_batch_invariance_and_spot_parity(assets, oracles, name, precision):
    """configs[1] / configs[2] at full size (256 x 1.5 s clips, PCM resident in HBM): a clip's posterior must
    not depend on what else is in the batch (bit-exact under chunking and permutation), and a sample of
    clips must match the oracle."""
    import torch
    from wwhip.engine import Engine, frontend_params
    e = Engine(os.path.join(assets, name), precision=precision)
    try:
        rng = np.random.default_rng(101)
        B, S = 256, 24000
        t = np.arange(S) / 16000.0
        chirp = 8000.0 * np.sin(2 * np.pi * (200.0 * t + 0.5 * 3800.0 / 1.5 * t * t))
        pcm = np.clip(np.rint(rng.normal(0, 2000, (B, S)) + chirp), -32768, 32767).astype(np.int16)
        fp = frontend_params()

        def run(batch):
            d_pcm = torch.from_numpy(np.ascontiguousarray(batch)).cuda()
            d_out = torch.zeros((len(batch), e.n_out), dtype=torch.float32, device="cuda")
            torch.cuda.synchronize()
            e.clips_forward_dev(d_pcm.data_ptr(), len(batch), S, d_out.data_ptr(), fp)
            e.ctx.synchronize()
            return d_out.cpu().numpy()

        full = run(pcm)
        assert np.isfinite(full).all() and (full >= 0).all() and (full <= 1).all()
        chunks = np.concatenate([run(pcm[i:i + 64]) for i in range(0, B, 64)])
        np.testing.assert_array_equal(full, chunks)
        perm = rng.permutation(B)
        np.testing.assert_array_equal(run(pcm[perm]), full[perm])
        np.testing.assert_array_equal(run(pcm), full)      # and it is deterministic run to run
        tol = TOL_POST if precision == "fp32" else 2e-5
        for i in rng.choice(B, 6, replace=False):
            mel = oracles[name].logmel(pcm[i])
            win = np.zeros((e.window, 40), np.float32)
            win[: min(len(mel), e.window)] = mel[: e.window]
            assert np.abs(full[i] - oracles[name].forward(win)[0]).max() < tol
    finally:
        e.close()


def test_logmel_parameter_sweep_vs_oracle(engines, oracles):
    """Both staging paths of the front end (straight-line: no pre-emphasis, divisor 32767/32768; generic:
    everything else), several hops, ragged lengths incl. ones that end inside a 16-byte vector."""
    from wwhip.engine import frontend_params
    rng = np.random.default_rng(57)
    e, o = engines["CRNN"], oracles["CRNN"]
    lens = [512, 527, 1000, 5003, 24000, 7, 24001, 16384]
    pcm = [np.clip(rng.normal(0, 4000, n), -32768, 32767).astype(np.int16) for n in lens]
    pcm[3][:40] = 32767      # clipping region for divisor 32767 is a no-op, for 30000 it is not
    cases = [(32767.0, True, 0.0, 160), (32768.0, False, 0.0, 160), (32767.0, True, 0.97, 160), (30000.0, True, 0.0, 160),
             (32768.0, False, 0.5, 80), (32767.0, False, 0.0, 200), (32767.0, True, 0.0, 512), (1000.0, True, 0.3, 37)]
    for div, clip, pre, hop in cases:
        got = e.logmel(pcm, frontend_params(div, clip, pre, hop, True))
        for n, p, g in zip(lens, pcm, got):
            want = o.logmel(p, divisor=div, clip=clip, preemph=pre, hop=hop)
            assert g.shape == want.shape, (div, clip, pre, hop, n)
            if len(want):
                assert np.abs(g - want).max() < 2e-5, (div, clip, pre, hop, n, float(np.abs(g - want).max()))
    with pytest.raises(ValueError):
        e.logmel(pcm, frontend_params(32767.0, True, 0.0, 0, True))
    with pytest.raises(ValueError):
        e.logmel(pcm, frontend_params(32767.0, True, 0.0, 513, True))


@pytest.mark.parametrize("name", ["CRNN", "Wavenet"])
def test_stream_bank_at_config5_width(engines, oracles, name):
    """BASELINE config 5 geometry on one GPU: 128 streams in lock step (every CU busy), is_speech = 1 on every stream (the
    configuration's worst case: two posteriors per stream and tick); a sample of streams is checked posterior by posterior
    against the batch oracle, and two runs must agree.  Mixed VAD bits, active stretches and resets at this width:
    test_stream_bank_mixed_state_at_width_vs_oracle below."""
    from wwhip.engine import StreamBank
    rng = np.random.default_rng(43)
    e = engines[name]
    S, ticks = 128, 100
    pcm = np.clip(rng.normal(0, 2500, (S, ticks * 320)), -32768, 32767).astype(np.int16)

    def run():
        bank = StreamBank(e, S)
        posts = [[] for _ in range(S)]
        speech = np.ones(S, np.uint8)
        for t in range(ticks):
            p, n = bank.step(pcm[:, t * 320:(t + 1) * 320], speech)
            for s in range(S):
                posts[s] += [float(p[s, k]) for k in range(n[s])]
        bank.close()
        return posts

    a, b = run(), run()
    assert a == b
    pidx = e.posterior_index
    for s in rng.choice(S, 5, replace=False):
        mel = oracles[name].logmel(pcm[s])
        hist = np.concatenate([np.zeros((e.window, 40), np.float32), mel])
        want = oracles[name].slide_forward(hist, 1)[1:, pidx]
        assert len(a[s]) == len(want)
        assert np.abs(np.array(a[s]) - want).max() < TOL_POST


@pytest.mark.parametrize("name", ["CRNN", "Wavenet"])
def test_stream_bank_mixed_state_at_width_vs_oracle(engines, oracles, name):
    """The default tick (ONE launch, polled posteriors) at BASELINE config 5's width with everything that makes streams differ:
    128 streams x 160 ticks, per-stream random VAD runs, two active stretches per stream (an active stream is not sampled: its
    rings stand still), WakewordTrigger.reset of single streams on every VAD fall and of the whole bank once, pre-emphasis 0.97
    (the carry survives resets).  Against the ORACLE, not against another launch form: n_post of every stream and tick against
    the closed form of the reference's sample loop, and the posteriors of 16 sampled streams one by one against the reference's
    gating restated (oracle/numpy_ref.RefGatedStream, spokestack/wakeword/tflite.py:123-168,241-246) feeding the C oracle's
    front end and model.  The single-writer / sibling logic of the one-launch kernels is exactly what differs between streams
    that gain 0, 1 or 2 frames, sit still, or restart in the same tick."""
    from oracle.numpy_ref import RefGatedStream
    from wwhip.engine import StreamBank, frontend_params
    rng = np.random.default_rng(2026)
    e, o = engines[name], oracles[name]
    S, TICKS, PRE, BANK_RESET_AT = 128, 160, 0.97, 101
    pcm = np.clip(rng.normal(0, 2500, (TICKS, S, 320)), -32768, 32767).astype(np.int16)
    speech = np.zeros((TICKS, S), bool)
    active = np.zeros((TICKS, S), bool)
    for s in range(S):
        t, v = 0, bool(rng.random() < 0.6)
        while t < TICKS:
            n = int(rng.integers(3, 30)) if v else int(rng.integers(1, 9))
            speech[t:t + n, s] = v
            t, v = t + n, not v
        for _ in range(2):
            a = int(rng.integers(5, TICKS - 25))
            active[a:a + int(rng.integers(3, 20)), s] = True
    bank = StreamBank(e, S, frontend_params(32767.0, True, PRE, 160, True))
    got = [[] for _ in range(S)]
    counts = np.zeros((TICKS, S), np.int32)
    try:
        for t in range(TICKS):
            if t == BANK_RESET_AT:
                bank.reset()
            p, n = bank.step(pcm[t], speech[t].astype(np.uint8), active[t].astype(np.uint8))
            counts[t] = n
            for s in np.nonzero(n)[0]:
                got[s] += [float(p[s, k]) for k in range(n[s])]
            fell = np.nonzero(speech[t - 1] & ~speech[t])[0] if t else np.zeros(0, np.int64)
            if len(fell):
                bank.reset(fell)  # tflite.py:143-146
    finally:
        bank.close()
    # ---- counts, all streams: f samples in the ring; a tick adds 320 and emits a frame per 160 while >= 512 are buffered
    fill = np.zeros(S, np.int64)
    for t in range(TICKS):
        if t == BANK_RESET_AT:
            fill[:] = 0
        live = ~active[t]
        tot = fill + 320
        nf = np.where(live & (tot >= 512), (tot - 512) // 160 + 1, 0)
        assert np.array_equal(counts[t], np.where(speech[t], nf, 0)), t
        fill = np.where(live, tot - 160 * nf, fill)
        if t:
            fill[speech[t - 1] & ~speech[t]] = 0
    assert {0, 1, 2} <= set(np.unique(counts)) and counts.sum() > 10000
    # ---- posteriors, 16 sampled streams: the reference's gating feeding the oracle
    pidx = e.posterior_index
    worst = 0.0
    for s in rng.choice(S, 16, replace=False):
        ref = RefGatedStream(lambda fr: o.logmel_f32(fr, 0.0, 160)[0], e.window, 40, pre_emphasis=PRE)
        wins = []
        for t in range(TICKS):
            if t == BANK_RESET_AT:
                ref.reset()
            w = ref.tick(pcm[t, s], bool(speech[t, s]), bool(active[t, s]))
            assert len(w) == counts[t, s], (s, t)
            wins += w
        assert len(wins) == len(got[s]) > 50, (s, len(wins))
        want = o.forward(np.array(wins))[:, pidx]
        worst = max(worst, float(np.abs(np.array(got[s]) - want).max()))
    assert worst < TOL_POST, worst


def test_context_adopts_a_torch_stream(assets, oracles):
    """ww_ctx_create(device, stream): work enqueued on a caller-owned HIP stream (here torch's) is ordered with
    the caller's own work on that stream - no explicit ww_ctx_synchronize needed before torch reads the result."""
    import torch
    from wwhip import _lib
    from wwhip.engine import Engine, frontend_params
    side = torch.cuda.Stream()
    ctx = _lib.Context(0, stream=side.cuda_stream)
    assert ctx.stream == side.cuda_stream
    e = Engine(os.path.join(assets, "CRNN"), ctx=ctx)
    try:
        rng = np.random.default_rng(61)
        pcm = np.clip(rng.normal(0, 2500, (16, 24000)), -32768, 32767).astype(np.int16)
        with torch.cuda.stream(side):
            d_pcm = torch.from_numpy(pcm).cuda(non_blocking=False)
            d_out = torch.zeros((16, e.n_out), dtype=torch.float32, device="cuda")
            e.clips_forward_dev(d_pcm.data_ptr(), 16, 24000, d_out.data_ptr(), frontend_params())
            doubled = d_out * 2.0            # torch kernel on the same stream, after ours
        side.synchronize()
        got = d_out.cpu().numpy()
        np.testing.assert_array_equal(doubled.cpu().numpy(), got * 2.0)
        for i in (0, 7, 15):
            mel = oracles["CRNN"].logmel(pcm[i])
            win = np.zeros((e.window, 40), np.float32)
            win[: len(mel)] = mel
            assert np.abs(got[i] - oracles["CRNN"].forward(win)[0]).max() < TOL_POST
    finally:
        e.close()
        ctx.close()


def test_stream_bank_on_a_borrowed_stream_and_its_timeline(assets):
    """A bank whose context BORROWS its stream (here torch's) keeps the old contract - when ww_stream_step returns, the stream
    has drained - so it waits with hipStreamSynchronize, never by polling; same posteriors as a bank on a stream of the
    library's own, one launch per tick in both.  And ww_stream_timeline: per-tick means of the six host phases, launch 2 is
    nothing in the one-launch form, the count resets."""
    import torch
    from wwhip import _lib
    from wwhip.engine import Engine, StreamBank
    side = torch.cuda.Stream()
    ctx = _lib.Context(0, stream=side.cuda_stream)
    e_own, e_bor = Engine(os.path.join(assets, "CRNN")), Engine(os.path.join(assets, "CRNN"), ctx=ctx)
    S, ticks = 5, 30
    rng = np.random.default_rng(17)
    pcm = np.clip(rng.normal(0, 2500, (ticks, S, 320)), -32768, 32767).astype(np.int16)
    speech = np.ones(S, np.uint8)
    try:
        own, bor = StreamBank(e_own, S), StreamBank(e_bor, S)
        for t in range(ticks):
            (p0, n0), (p1, n1) = own.step(pcm[t], speech), bor.step(pcm[t], speech)
            np.testing.assert_array_equal(n0, n1)
            np.testing.assert_array_equal(p0, p1)
            assert side.query()                      # nothing of the tick is still in flight on the borrowed stream
        for bank in (own, bor):
            tl = bank.timeline()
            assert tl["ticks"] == ticks and tl["launch_2"] < 0.5 and tl["launch_1"] > 0.5 and tl["wait"] > 5.0
            assert set(tl) == set(StreamBank.TIMELINE_PHASES) | {"ticks"}
            assert bank.timeline(reset=True)["ticks"] == ticks and bank.timeline()["ticks"] == 0
        two = StreamBank(e_own, S, two_launch=True)
        two.step(pcm[0], speech)
        assert two.timeline()["launch_2"] < 0.2      # (no window in a bank's first tick: the front end alone)
        two.step(pcm[1], speech)
        two.step(pcm[2], speech)
        assert two.timeline()["launch_2"] > 0.3      # front-end kernel + model kernel
        for bank in (own, bor, two):
            bank.close()
    finally:
        e_own.close()
        e_bor.close()
        ctx.close()


def _engine_with_filter(assets, monkeypatch, weight):
    """A CRNN engine whose filter.tflite weights are replaced by ``weight`` [40][257]."""
    import dataclasses
    from wwhip import weights as W
    from wwhip.engine import Engine
    real = W.load_model_dir

    def patched(path):
        b = real(path)
        return dataclasses.replace(b, filt=dataclasses.replace(b.filt, weight=np.ascontiguousarray(weight, np.float32)))

    monkeypatch.setattr(W, "load_model_dir", patched)
    return Engine(os.path.join(assets, "CRNN"))


def test_logmel_with_other_filterbanks(assets, monkeypatch):
    """The lane form of the mel filter (api.hip load_filter) on filterbanks other than the shipped one: a bank
    whose widest band cannot be moved to a 16-byte boundary (scalar-read path), bands without any weight, bands of
    one tap, bands at the very top of the spectrum - and a band wider than the form takes (load error)."""
    from oracle.cpu import CpuOracle
    rng = np.random.default_rng(91)
    pcm = [np.clip(rng.normal(0, 3000, n), -32768, 32767).astype(np.int16) for n in (24000, 3000, 700)]

    def bank(spans):
        w = np.zeros((40, 257), np.float32)
        for b, (s, n) in enumerate(spans):
            w[b, s:s + n] = rng.uniform(0.05, 1.0, n).astype(np.float32)
        return w

    banks = {
        # widest band 36 taps starting on an odd bin: no room to round its first bin down
        "unaligned": bank([(217, 36), (181, 34), (150, 31)] + [(3 * i + 1, 5 + i % 7) for i in range(37)]),
        # dead bands, one-tap bands, bands touching bin 256
        "sparse": bank([(256, 1), (250, 7), (245, 12), (0, 1), (1, 1)] + [(0, 0)] * 5 + [(5 * i, 3) for i in range(30)]),
        # sixteen wide bands, sixteen of 16 taps, eight of 12: every group at its limit
        "full": bank([(7 * i + 1, 36) for i in range(16)] + [(11 * i + 2, 16) for i in range(16)] + [(30 * i + 3, 12) for i in range(8)]),
    }
    for name, w in banks.items():
        e = _engine_with_filter(assets, monkeypatch, w)
        try:
            o = CpuOracle(e.blob)
            for fast in (False, True):
                from wwhip.engine import frontend_params
                got = e.logmel(pcm, frontend_params(precise=not fast))
                for p, g in zip(pcm, got):
                    want = o.logmel(p)
                    assert g.shape == want.shape
                    assert np.abs(g - want).max() < (2e-3 if fast else 2e-5), (name, fast, float(np.abs(g - want).max()))
        finally:
            e.close()
    with pytest.raises(Exception, match="mel band"):
        _engine_with_filter(assets, monkeypatch, bank([(10, 37)] + [(4 * i, 4) for i in range(39)]))
    with pytest.raises(Exception, match="mel band"):
        _engine_with_filter(assets, monkeypatch, bank([(7 * i, 20) for i in range(17)] + [(4 * i, 4) for i in range(23)]))


@pytest.mark.parametrize("name", ["CRNN", "Wavenet"])
def test_degenerate_sizes_through_the_c_abi(engines, oracles, name):
    """Empty and minimal inputs: zero windows, a mel stream shorter than / exactly one window, empty posterior sets in the
    sweep - the entry points return empty results (status 0), not errors, and the one-window cases equal the oracle."""
    e = engines[name]
    rng = np.random.default_rng(41)
    out, enc = e.forward(np.zeros((0, e.window, 40), np.float32), want_enc=True)
    assert out.shape == (0, e.n_out) and enc.shape[0] == 0
    assert e.slide_forward(np.zeros((e.window - 1, 40), np.float32), 2).shape == (0, e.n_out)
    assert e.slide_forward(np.zeros((0, 40), np.float32), 2).shape == (0, e.n_out)
    mel = rng.uniform(0, 6.5, (e.window, 40)).astype(np.float32)
    one = e.slide_forward(mel, 2)
    assert one.shape == (1, e.n_out)
    assert np.abs(one - oracles[name].forward(mel[None])).max() < TOL_POST
    assert np.abs(e.slide_forward(np.concatenate([mel, mel[:1]]), 2) - one).max() == 0   # one extra row: still one window at hop 2
    assert e.logmel([]) == []
    thr = np.arange(0.5, 0.99999, 0.005)
    frr, fa, cnt = e.far_frr(np.zeros(0, np.float32), np.zeros(0, np.float32), thr, 1.0, 1.0)
    assert frr.shape == fa.shape == cnt.shape == thr.shape and not cnt.any() and not fa.any()
    frr, fa, cnt = e.far_frr(np.array([0.9], np.float32), np.zeros(0, np.float32), thr, 1.0, 1.0)
    assert frr[0] == 0.0 and frr[-1] == 1.0 and not fa.any()   # 0.9 > 0.5, 0.9 < 0.995


@pytest.mark.parametrize("name", ["CRNN", "CRNN_softmax"])
def test_stream_incremental_kernel_matches_full_recompute(assets, name):
    """Streaming CRNN: crnn_stream_kernel (three of the nineteen time positions per new window, the other sixteen projected
    rows from the per-stream ring) against the full recompute of every window (a bank created with
    WW_STREAM_FULL_RECOMPUTE) - 300 ticks with
    the speech bit going on and off, resets of single streams and of the whole bank, from an empty history."""
    from wwhip.engine import Engine, StreamBank
    e = Engine(os.path.join(assets, name))
    S, ticks = 7, 300
    rng = np.random.default_rng(77)
    pcm = np.clip(rng.normal(0, 2500, (ticks, S, 320)), -32768, 32767).astype(np.int16)
    speech = (rng.uniform(size=(ticks, S)) < 0.9).astype(np.uint8)
    speech[:, 0] = 1

    def run(full):
        bank = StreamBank(e, S, full_recompute=full)
        out = []
        for t in range(ticks):
            if t == 120:
                bank.reset([2, 5])
            if t == 200:
                bank.reset()
            p, n = bank.step(pcm[t], speech[t])
            out.append((p.copy(), n.copy()))
        bank.close()
        return out

    try:
        inc, full = run(False), run(True)
        worst = 0.0
        for (p0, n0), (p1, n1) in zip(inc, full):
            np.testing.assert_array_equal(n0, n1)
            worst = max(worst, float(np.abs(p0 - p1).max()))
        assert worst < 2e-6, worst   # the two kernels associate the projection's k sum differently for some rows
        assert sum(int(n.sum()) for _, n in inc) > 3000
    finally:
        e.close()


def test_posterior_pick_and_sweep_on_the_device(assets):
    """a13 / a14 / a15 / a16 without leaving the device (round 5: what the sharded evaluators keep of a chunk and how they
    finish): ww_posterior_pick_dev = element [posterior_index] of detect rows, as they are or as the maximum of each run of
    windows (utils/evaluate_models.py:80,98-99) - bit for bit against NumPy, runs of one window and ragged runs included; and
    ww_far_frr_dev over device pointers = ww_far_frr over host arrays = the reference's loop restated in oracle/numpy_ref.py."""
    import torch
    from wwhip.engine import Engine
    from oracle import numpy_ref as NR
    rng = np.random.default_rng(8)
    for name in ("CRNN_softmax", "CRNN"):
        e = Engine(os.path.join(assets, name))
        try:
            n = 50_000
            rows = rng.uniform(0, 1, (n, e.n_out)).astype(np.float32)
            lens = rng.integers(1, 120, 2000)
            lens = lens[np.cumsum(lens) <= n]
            offs = np.concatenate(([0], np.cumsum(lens))).astype(np.int64)
            d_rows, d_offs = torch.from_numpy(rows).cuda(), torch.from_numpy(offs).cuda()
            d_all, d_max = torch.empty(n, device="cuda"), torch.empty(len(lens), device="cuda")
            torch.cuda.synchronize()
            e.posterior_pick_dev(d_rows.data_ptr(), n, d_all.data_ptr())
            e.posterior_pick_dev(d_rows.data_ptr(), n, d_max.data_ptr(), d_offs.data_ptr(), len(lens))
            e.ctx.synchronize()
            col = rows[:, e.posterior_index]
            np.testing.assert_array_equal(d_all.cpu().numpy(), col)
            np.testing.assert_array_equal(d_max.cpu().numpy(), np.maximum.reduceat(col, offs[:-1]))
            # the sweep over device pointers
            pos = rng.uniform(0.3, 1, 700).astype(np.float32)
            neg = np.clip(rng.normal(0.45, 0.2, 12_000), 0, 1).astype(np.float32)
            thr = np.arange(0.3, 0.99999, 0.005)
            d_pos, d_neg = torch.from_numpy(pos).cuda(), torch.from_numpy(neg).cuda()
            torch.cuda.synchronize()
            got = e.far_frr_dev(d_pos.data_ptr(), len(pos), d_neg.data_ptr(), len(neg), thr, 700.0, 1.25)
            want = e.far_frr(pos, neg, thr, 700.0, 1.25)
            ref = NR.far_frr(pos, neg, 700, 1.25, thr)
            for g, w in zip(got, want):
                np.testing.assert_array_equal(g, w)
            np.testing.assert_array_equal(got[2], ref[2])   # FA counts: the reference's loop
            assert got[2].max() > 50
        finally:
            e.close()


@pytest.mark.parametrize("name,precise,prec", [("CRNN", True, "fp32"), ("CRNN_softmax", True, "fp32"), ("CRNN", False, "fp32"),
                                               ("Wavenet", True, "fp32"), ("Wavenet", True, "bf16x3"), ("Wavenet_alt", False, "fp32")])
def test_one_launch_tick_equals_the_two_launch_form(assets, name, precise, prec):
    """Round 5: a tick of the incremental CRNN bank - and of the Wavenet banks - is ONE launch: the front end of a stream's new
    frames runs inside the workgroups of that stream's new windows, the workgroup of the newest window alone writes the stream's
    state (sample ring and carry ping-pong by the stream's parity so that its sibling still reads last tick's) - and the host polls the
    posteriors' {value, tick number} pairs.  Against the front-end kernel + model kernel form waited for with
    hipStreamSynchronize: 320 ticks with the VAD bit going on and off per stream (a silent stream's ring still advances; a
    whole tick without a window), streams that are active for a while (not sampled at all: tflite.py:139-140), resets of
    single streams and of the whole bank, pre-emphasis on - bit for bit, every tick."""
    from wwhip.engine import Engine, StreamBank, frontend_params
    e = Engine(os.path.join(assets, name), precision=prec)
    S, ticks = 9, 320
    rng = np.random.default_rng(505)
    pcm = np.clip(rng.normal(0, 2500, (ticks, S, 320)), -32768, 32767).astype(np.int16)
    speech = (rng.uniform(size=(ticks, S)) < 0.85).astype(np.uint8)
    speech[:, 0] = 1
    speech[60:64] = 0                       # whole ticks without a window (the host then waits for the stream instead)
    speech[150:190, 3] = 0                  # a long silence: the ring advances, the mel history stands still
    active = np.zeros((ticks, S), np.uint8)
    active[90:140, 4] = 1                   # an activated stream is not sampled until the flag is cleared
    active[:, 7] = rng.uniform(size=ticks) < 0.2
    fp = frontend_params(pre_emphasis=0.97, precise=precise)

    def run(**kw):
        bank = StreamBank(e, S, fp=fp, **kw)
        out = []
        for t in range(ticks):
            if t == 120:
                bank.reset([2, 5])
            if t == 200:
                bank.reset()
            out.append(bank.step(pcm[t], speech[t], active[t]))
        bank.close()
        return out

    try:
        one = run()
        two = run(two_launch=True, sync_wait=True)
        for t, ((p0, n0), (p1, n1)) in enumerate(zip(one, two)):
            np.testing.assert_array_equal(n0, n1, err_msg=f"tick {t}")
            np.testing.assert_array_equal(p0, p1, err_msg=f"tick {t}")
        assert sum(int(n.sum()) for _, n in one) > 3500
        assert any(int(n.sum()) == 0 for _, n in one[10:])
        if not e.is_crnn:  # the fp32 Wavenet's row-major block loop (an option of the model) in both tick forms
            with e.options(wavenet_rowmajor=1):
                for (p0, n0), (p1, n1) in zip(run(), run(two_launch=True, sync_wait=True)):
                    np.testing.assert_array_equal(p0, p1)
        for kw in ({"two_launch": True}, {"sync_wait": True}):   # the other two combinations: polled two-launch, waited one-launch
            other = run(**kw)
            for t, ((p0, n0), (p1, n1)) in enumerate(zip(one, other)):
                np.testing.assert_array_equal(p0, p1, err_msg=f"{kw} tick {t}")
    finally:
        e.close()


@pytest.mark.parametrize("name,S", [("CRNN_softmax", 700), ("Wavenet", 128), ("Wavenet", 300)])
def test_one_launch_tick_with_many_streams_per_gpu(assets, name, S):
    """The one-launch tick past one workgroup per CU: 700 CRNN streams = 1,400 workgroups (several co-resident per CU, siblings
    of one stream anywhere on the chip), every third stream silent, against the two-launch form - bit for bit - over 14 ticks
    from an empty history (the first ticks yield 0, 1, 2 windows per stream).  Wavenet banks take the one-launch form up to 128
    streams (a tick's 256 windows are the twelve-wave kernel's range) and the two-launch form with the wide kernel above."""
    from wwhip.engine import Engine, StreamBank
    e = Engine(os.path.join(assets, name))
    ticks = 14
    rng = np.random.default_rng(3)
    pcm = np.clip(rng.normal(0, 2500, (ticks, S, 320)), -32768, 32767).astype(np.int16)
    speech = np.ones(S, np.uint8)
    speech[::3] = 0
    try:
        outs = []
        for kw in ({}, {"two_launch": True, "sync_wait": True}):
            bank = StreamBank(e, S, **kw)
            outs.append([bank.step(pcm[t], speech) for t in range(ticks)])
            bank.close()
        for (p0, n0), (p1, n1) in zip(*outs):
            np.testing.assert_array_equal(n0, n1)
            np.testing.assert_array_equal(p0, p1)
        assert sum(int(n.sum()) for _, n in outs[0]) > 14 * S
        tl = StreamBank(e, S)
        for t in range(4):
            tl.step(pcm[t], speech)
        assert (tl.timeline()["launch_2"] < 0.3) == (e.is_crnn or S <= 128)   # which form the bank took
        tl.close()
    finally:
        e.close()


@pytest.mark.parametrize("name,prec", [("Wavenet", "fp32"), ("Wavenet", "bf16x3"), ("CRNN", "bf16x3")])
def test_polled_tick_equals_the_waited_one(assets, name, prec):
    """Every bank whose context owns its stream waits for a tick by polling the {value, tick number} pairs the heads store into
    page-locked memory (wavenet_kernel, the one-kernel CRNN forms); WW_STREAM_SYNC_WAIT keeps hipStreamSynchronize and rows of
    h_out.  Same posteriors."""
    from wwhip.engine import Engine, StreamBank
    e = Engine(os.path.join(assets, name), precision=prec)
    S, ticks = 6, 40
    rng = np.random.default_rng(11)
    pcm = np.clip(rng.normal(0, 2500, (ticks, S, 320)), -32768, 32767).astype(np.int16)
    speech = (rng.uniform(size=(ticks, S)) < 0.8).astype(np.uint8)
    try:
        outs = []
        ro = pcm.copy()
        ro.setflags(write=False)  # (the wrapper takes a frame array's address without a ctypes round trip; a read-only one the slow way)
        for kw in ({}, {"sync_wait": True}, {"full_recompute": True}, {"full_recompute": True, "sync_wait": True}):
            bank = StreamBank(e, S, **kw)
            if kw:
                outs.append([bank.step(pcm[t], speech[t]) for t in range(ticks)])
            else:  # the same ticks from a read-only array, a Fortran-ordered one, plain lists and booleans
                forms = (lambda t: ro[t], lambda t: np.asfortranarray(pcm[t]), lambda t: pcm[t].tolist(), lambda t: pcm[t])
                outs.append([bank.step(forms[t % 4](t), speech[t].astype(bool) if t % 3 == 0 else speech[t].tolist() if t % 3 == 1 else speech[t])
                             for t in range(ticks)])
            with pytest.raises(ValueError):
                bank.step(pcm[0][:-1], speech[0])
            bank.close()
        for a, b in ((0, 1), (2, 3)):
            for (p0, n0), (p1, n1) in zip(outs[a], outs[b]):
                np.testing.assert_array_equal(n0, n1)
                np.testing.assert_array_equal(p0, p1)
        assert sum(int(n.sum()) for _, n in outs[0]) > 250
    finally:
        e.close()


def test_full_recompute_bank_survives_an_option_change_after_its_creation(assets):
    """A WW_STREAM_FULL_RECOMPUTE bank sizes its model scratch when it is created (1 KB while 2 S <= crnn_split_at: the fused
    kernel keeps everything in LDS).  Options are per model and mutable: lowering the front/tail threshold afterwards sends
    the bank's ticks to crnn_fused_kernel<front> + a tail, which want 2 S x 19 x 192 floats of scratch - the bank must grow
    its buffer (round 3 wrote past the 1 KB one), and since every form rounds alike the posteriors must not change at all."""
    from wwhip.engine import Engine, StreamBank
    e = Engine(os.path.join(assets, "CRNN_softmax"))
    S, ticks = 100, 24
    rng = np.random.default_rng(5)
    pcm = np.clip(rng.normal(0, 2500, (ticks, S, 320)), -32768, 32767).astype(np.int16)
    speech = np.ones(S, np.uint8)
    try:
        ref_bank, bank = StreamBank(e, S, full_recompute=True), StreamBank(e, S, full_recompute=True)
        want = [ref_bank.step(pcm[t], speech)[0].copy() for t in range(ticks)]
        got = []
        for t in range(ticks):
            if t == 8:
                e.set_option("crnn_split_at", 64)       # 200 windows per tick > 64: front + tail from here on
            if t == 16:
                e.set_option("crnn_tail_mfma", 2)       # ... and the sixteen-window tail, which wants its seq scratch too
            got.append(bank.step(pcm[t], speech)[0].copy())
        for t in range(ticks):
            np.testing.assert_array_equal(got[t], want[t])
        assert np.abs(np.array(want)).max() > 0
        ref_bank.close()
        bank.close()
    finally:
        e.close()


@pytest.mark.parametrize("name", ["CRNN", "CRNN_softmax"])
def test_crnn_sliding_rows_path_matches_per_window_kernels(engines, oracles, name):
    """From 64 regular sliding windows on, the CRNN computes every projected row once per sequence (crnn_rows_kernel: one new
    interior field and the two edge positions per window) and gru_tail_kernel gathers a window's 19 rows.  Same posteriors
    as the per-window kernels on the same windows (stacked: hop = T takes crnn_fused_kernel), for every hop class
    (gcd with 8 = 1, 2, 4, 8), a sequence that ends exactly with a window, and a chunk boundary."""
    e = engines[name]
    rng = np.random.default_rng(97)
    T = e.window
    for hop, nw in ((2, 300), (1, 130), (3, 77), (4, 64), (8, 100), (5, 65), (2, 1500)):
        rows = (nw - 1) * hop + T + (0 if hop == 4 else 3)
        mel = rng.uniform(0, 6.5, (rows, 40)).astype(np.float32)
        mel[rng.integers(0, rows, 5)] = 0
        with e.options(crnn_tail_mfma=2):
            got = e.slide_forward(mel, hop)                 # gathering gru_tail16_kernel (16 windows per workgroup, MFMA)
        with e.options(crnn_tail_mfma=0):
            got_valu = e.slide_forward(mel, hop)            # gathering gru_tail_kernel (one window per workgroup, vector ALU)
        np.testing.assert_array_equal(got, got_valu)                         # one arithmetic, whichever tail (round 4)
        np.testing.assert_array_equal(e.slide_forward(mel, hop), got_valu)   # default: the matrix form only from 9,216 windows on
        assert got.shape[0] == (rows - T) // hop + 1 >= nw
        wins = np.stack([mel[i * hop:i * hop + T] for i in range(got.shape[0])])
        ref = np.concatenate([e.forward(wins[i:i + 512]) for i in range(0, len(wins), 512)])
        assert np.abs(got - ref).max() < 2e-6, (hop, nw, float(np.abs(got - ref).max()))
        assert np.abs(got_valu - ref).max() < 2e-6, (hop, nw, float(np.abs(got_valu - ref).max()))
        idx = rng.choice(len(wins), 24, replace=False)
        assert np.abs(got[idx] - oracles[name].forward(wins[idx])).max() < TOL_POST


def test_wavenet_split_bf16_launch_forms_agree(assets):
    """Split-bf16 Wavenet: launches of up to 256 windows run twelve waves x one tile per window, larger ones four waves x
    three tiles with two workgroups per CU (wavenet.hip, ww_k_wave_forward).  The per-tile arithmetic is the same: 700 windows
    in one launch (wide form), in launches of 200 (one tile per wave) and of 300 (wide) give the same bits, encoder output
    included; windows of partial validity among them."""
    from wwhip.engine import Engine
    e = Engine(os.path.join(assets, "Wavenet"), precision="bf16x3")
    try:
        rng = np.random.default_rng(41)
        wins = rng.uniform(0, 6.5, (700, e.window, 40)).astype(np.float32)
        wins[::9, 150:] = 0
        whole, whole_enc = e.forward(wins, want_enc=True)
        for step in (200, 300):
            parts = [e.forward(wins[i:i + step], want_enc=True) for i in range(0, 700, step)]
            np.testing.assert_array_equal(np.concatenate([p[0] for p in parts]), whole)
            np.testing.assert_array_equal(np.concatenate([p[1] for p in parts]), whole_enc)
    finally:
        e.close()


@pytest.mark.parametrize("name", ["Wavenet", "Wavenet_alt"])
def test_wavenet_fp32_launch_forms_agree(assets, oracles, name):
    """fp32 Wavenet (round 5): the transposed block loop is written over a wave's tiles like the split-bf16 one, so launches of
    more than 256 windows run four waves x three tiles with two workgroups per CU, smaller ones twelve waves x one tile.  The
    per-tile arithmetic is one source: 700 windows in one launch (wide), in launches of 200 (one tile per wave) and of 300
    (wide) give the same bits, encoder output included, windows of partial validity among them - and the oracle's posteriors."""
    from wwhip.engine import Engine
    e = Engine(os.path.join(assets, name))
    try:
        rng = np.random.default_rng(43)
        wins = rng.uniform(0, 6.5, (700, e.window, 40)).astype(np.float32)
        wins[::9, 150:] = 0
        wins[5] = 0
        whole, whole_enc = e.forward(wins, want_enc=True)
        for step in (200, 300):
            parts = [e.forward(wins[i:i + step], want_enc=True) for i in range(0, 700, step)]
            np.testing.assert_array_equal(np.concatenate([p[0] for p in parts]), whole)
            np.testing.assert_array_equal(np.concatenate([p[1] for p in parts]), whole_enc)
        idx = np.arange(0, 700, 37)
        assert np.abs(whole[idx] - oracles[name].forward(wins[idx])).max() < TOL_POST
    finally:
        e.close()


def test_crnn_default_dispatch_crossover_leaves_posteriors_alone(engines):
    """With the library's default options the tail kernel changes at 9,216 windows per launch (gru_tail_kernel below,
    gru_tail16_kernel from there on) and public calls chunk at 16,384: a posterior must not depend on the size of the call it
    was part of, nor on its position in it.  20,000 sliding windows in one call (a 16,384-window chunk on the matrix tail + a
    3,616-window chunk on the vector tail) against the same windows in calls of 5,000 (vector tail throughout) and with either
    tail forced: bit for bit."""
    e = engines["CRNN_softmax"]
    rng = np.random.default_rng(99)
    nw, hop, T = 20000, 2, e.window
    mel = rng.uniform(0, 6.5, ((nw - 1) * hop + T, 40)).astype(np.float32)
    whole = e.slide_forward(mel, hop)
    assert whole.shape[0] == nw
    parts = np.concatenate([e.slide_forward(mel[i * hop:(i + 5000 - 1) * hop + T], hop) for i in range(0, nw, 5000)])
    np.testing.assert_array_equal(whole, parts)
    for mfma in (0, 2):
        with e.options(crnn_tail_mfma=mfma):
            np.testing.assert_array_equal(e.slide_forward(mel, hop), whole)


@pytest.mark.parametrize("name", ["CRNN", "Wavenet"])
def test_forward_segments_dev_matches_explicit_windows(engines, oracles, name):
    """ww_forward_segments_dev (several sequences in one mel buffer, each slid over with the same hop) against the same
    windows given one by one - CRNN: crnn_rows_kernel by tile descriptors + the gathering gru_tail_kernel; Wavenet: the
    entry point's explicit-list fallback.  Sequences of 0, 1, 15, 16, 17 and many windows, odd and even first rows."""
    import torch
    e = engines[name]
    T = e.window
    rng = np.random.default_rng(123)
    for hop in (2, 3):
        seg_nw = np.array([40, 0, 1, 15, 16, 17, 333, 2], np.int32)
        gaps = np.array([0, 5, 3, 0, 7, 1, 2, 9])
        rows, r = [], 0
        for nw, gap in zip(seg_nw, gaps):
            r += int(gap)
            rows.append(r)
            r += ((int(nw) - 1) * hop + T) if nw else 11
        mel = rng.uniform(0, 6.5, (r + 4, 40)).astype(np.float32)
        seg_row0 = np.array(rows, np.int64)
        d_mel = torch.from_numpy(mel).cuda()
        n = int(seg_nw.sum())
        d_out = torch.zeros((n, e.n_out), dtype=torch.float32, device="cuda")
        torch.cuda.synchronize()
        e.forward_segments_dev(d_mel.data_ptr(), len(mel), seg_row0, seg_nw, hop, d_out.data_ptr())
        e.ctx.synchronize()
        got = d_out.cpu().numpy()
        wins = np.stack([mel[r0 + k * hop: r0 + k * hop + T] for r0, nw in zip(seg_row0, seg_nw) for k in range(nw)])
        ref = e.forward(wins)
        assert np.abs(got - ref).max() < 2e-6, (hop, float(np.abs(got - ref).max()))
        if e.is_crnn:  # the matrix tail (what the evaluation flows pin) with per-window first rows across sequence boundaries
            d_out.zero_()
            with e.options(crnn_tail_mfma=2):
                e.forward_segments_dev(d_mel.data_ptr(), len(mel), seg_row0, seg_nw, hop, d_out.data_ptr())
            e.ctx.synchronize()
            assert np.abs(d_out.cpu().numpy() - ref).max() < 2e-6
        idx = rng.choice(len(wins), 16, replace=False)  # and against the CPU oracle, not only the library's other kernels
        assert np.abs(got[idx] - oracles[name].forward(wins[idx])).max() < TOL_POST
    with pytest.raises(ValueError):
        e.forward_segments_dev(d_mel.data_ptr(), len(mel), np.array([len(mel) - 10], np.int64), np.array([3], np.int32), 2, d_out.data_ptr())


def test_two_host_threads_through_the_c_abi(assets, oracles):
    """include/wwhip.h: "one ww_ctx per host thread".  Two Python threads (ctypes releases the GIL inside every call, so
    the library really runs concurrently), each with its OWN context, model and stream bank - one CRNN, one Wavenet -
    interleave ww_logmel / ww_forward / ww_slide_forward / ww_stream_step calls; every result equals the same calls made
    from one thread, and the oracle.  A failed ww_ctx_create in each thread leaves its own message in that thread's
    ww_last_error(NULL) (thread-local text)."""
    import ctypes as C
    import threading
    from wwhip import _lib
    from wwhip.engine import Engine, StreamBank, frontend_params
    rounds = 12

    def job(name, seed, barrier=None):
        ctx = _lib.Context(0)
        eng = Engine(os.path.join(assets, name), ctx=ctx)
        bank = StreamBank(eng, 5)
        rng = np.random.default_rng(seed)
        out = []
        for r in range(rounds):
            if barrier is not None:
                barrier.wait()
            pcm = [np.clip(rng.normal(0, 2500, n), -32768, 32767).astype(np.int16) for n in (24000, 9000 + 37 * r)]
            mels = eng.logmel(pcm, frontend_params())
            win = np.zeros((3, eng.window, 40), np.float32)
            win[0, :min(len(mels[0]), eng.window)] = mels[0][:eng.window]
            win[1] = rng.uniform(0, 6.5, (eng.window, 40))
            post = eng.forward(win)
            slide = eng.slide_forward(rng.uniform(0, 6.5, (eng.window + 2 * (70 + r), 40)).astype(np.float32), 2)
            ticks = [bank.step(np.clip(rng.normal(0, 2500, (5, 320)), -32768, 32767).astype(np.int16), np.ones(5, np.uint8))
                     for _ in range(4)]
            out.append((mels, win, post, slide, ticks))
        bank.close()
        eng.close()
        ctx.close()
        return out

    want = {name: job(name, seed) for name, seed in (("CRNN", 41), ("Wavenet", 42))}   # one thread, one after the other
    got, errs, texts = {}, [], {}
    barrier = threading.Barrier(2)

    def worker(name, seed, bad_device):
        try:
            h = C.c_void_p()
            rc = _lib.load().ww_ctx_create(bad_device, None, C.byref(h))
            barrier.wait()                                                 # both creates have failed before either text is read
            texts[name] = (rc, _lib.load().ww_last_error(None).decode())
            got[name] = job(name, seed, barrier)
        except Exception as e:  # pragma: no cover - reported below
            errs.append((name, repr(e)))
            barrier.abort()

    th = [threading.Thread(target=worker, args=a) for a in (("CRNN", 41, 1234), ("Wavenet", 42, -7))]
    for t in th:
        t.start()
    for t in th:
        t.join(600)
    assert not errs, errs
    assert texts["CRNN"][0] == _lib.WW_EINVAL and "1234" in texts["CRNN"][1]
    assert texts["Wavenet"][0] == _lib.WW_EINVAL and "-7" in texts["Wavenet"][1]
    for name in ("CRNN", "Wavenet"):
        for (m0, w0, p0, s0, t0), (m1, w1, p1, s1, t1) in zip(want[name], got[name]):
            for a, b in zip(m0, m1):
                np.testing.assert_array_equal(a, b)
            np.testing.assert_array_equal(p0, p1)
            np.testing.assert_array_equal(s0, s1)
            for (pa, na), (pb, nb) in zip(t0, t1):
                np.testing.assert_array_equal(na, nb)
                np.testing.assert_array_equal(pa, pb)
        _, win, post, _, _ = got[name][-1]
        assert np.abs(post - oracles[name].forward(win)).max() < TOL_POST


def test_short_lived_threads_release_their_contexts_and_engines(assets):
    """The drop-in classes give every host thread its own ww_ctx and its own engines (wwhip/_lib.default_context,
    wwhip/models.engine_for).  They live in thread-local storage: a thread-per-request host must get the stream, arenas and
    uploaded weights of a finished thread back - six threads one after another leave no more live contexts / models than
    there were, and each computed the same posteriors."""
    import gc
    import threading
    from wwhip import _lib
    from wwhip.models import engine_for
    mdir = os.path.join(assets, "CRNN")
    wins = np.random.default_rng(8).uniform(0, 6.5, (5, 151, 40)).astype(np.float32)
    want = engine_for(mdir).forward(wins)
    gc.collect()
    base_ctx, base_models = len(_lib._live["contexts"]), len(_lib._live["models"])
    outs, errs = [], []

    def work():
        try:
            eng = engine_for(mdir)                       # this thread's engine on this thread's context
            assert eng.ctx is _lib.default_context(0)
            outs.append(eng.forward(wins))
        except Exception as e:                           # pragma: no cover
            errs.append(e)

    for _ in range(6):
        t = threading.Thread(target=work)
        t.start()
        t.join()
        gc.collect()
        assert len(_lib._live["contexts"]) <= base_ctx + 1 and len(_lib._live["models"]) <= base_models + 1
    assert not errs and len(outs) == 6
    for o in outs:
        np.testing.assert_array_equal(o, want)
    gc.collect()
    assert len(_lib._live["contexts"]) == base_ctx and len(_lib._live["models"]) == base_models


@pytest.mark.parametrize("name", ["Wavenet", "Wavenet_alt"])
def test_wavenet_fp32_block_loop_forms(engines, oracles, name):
    """The fp32 Wavenet's block loop in its two forms: transposed (default since round 3: channels x time, BatchNorm output and
    gate product feed the next MFMA from registers) and row-major (option wavenet_rowmajor = 1, rounds 1-2).  Same products,
    another summation order of the taps and the bias: both within tolerance of the oracle, within 2e-6 of each other, for
    full, partially valid and all-zero windows, encoder output included."""
    e = engines[name]
    rng = np.random.default_rng(55)
    wins = rng.uniform(0, 6.5, (70, e.window, 40)).astype(np.float32)
    wins[1] = 0
    wins[2, 100:] = 0
    wins[3] = np.clip(rng.normal(3, 1.5, (e.window, 40)), 0, 8)
    want, want_enc = oracles[name].forward(wins, want_enc=True)
    got_t, enc_t = e.forward(wins, want_enc=True)
    with e.options(wavenet_rowmajor=1):
        got_r, enc_r = e.forward(wins, want_enc=True)
    for got, enc in ((got_t, enc_t), (got_r, enc_r)):
        assert np.abs(got - want).max() < TOL_POST
        assert np.abs(enc.reshape(want_enc.shape) - want_enc).max() < 1e-4
    assert np.abs(got_t - got_r).max() < 2e-6
    assert np.abs(enc_t - enc_r).max() < 2e-5
    mel = rng.uniform(0, 6.5, (e.window + 60, 40)).astype(np.float32)
    assert np.abs(e.slide_forward(mel, 2) - oracles[name].slide_forward(mel, 2)).max() < TOL_POST


@pytest.mark.timeout(300)
def test_uploader_delivers_what_host_staging_writes(assets):
    """ww_uploader_* (include/wwhip.h): the library thread that assembles a chunk of sample runs in a page-locked slot and
    sends it to the device, against ww_host_stage_i16 into ordinary memory on the same runs - twelve chunks through three
    slots (every slot reused four times, sizes growing and shrinking so the slots are re-allocated on the way), runs that
    start at odd sample offsets, gaps of every size, a chunk with no runs at all (all zeros) and an empty table; tickets count
    1, 2, ...; poll turns 1; a chunk whose runs overlap is refused at its wait (ValueError) without upsetting its
    neighbours; a ticket is waited for once; an uploader serves only contexts of its device; destroy finishes what was
    submitted."""
    import torch
    from wwhip import _lib
    lib = _lib.load()
    ctx = _lib.default_context(0)
    up = _lib.Uploader(ctx, slots=3, copy_threads=4)
    rng = np.random.default_rng(11)
    clips = [rng.integers(-30000, 30000, int(n)).astype(np.int16) for n in rng.integers(50, 90000, 40)]
    jobs = []
    for k in range(12):
        total = int(rng.integers(1, 5_000_000)) if k not in (3, 7) else (2_500_000 if k == 3 else 300)
        n_runs = 0 if k == 5 else int(rng.integers(1, 30))
        # ascending disjoint runs inside [0, total)
        cuts = np.sort(rng.choice(total + 1, 2 * n_runs, replace=total + 1 < 2 * n_runs)) if n_runs else np.zeros(0, np.int64)
        d, p, c, keep = [], [], [], []
        for j in range(n_runs):
            lo, hi = int(cuts[2 * j]), int(cuts[2 * j + 1])
            src = clips[int(rng.integers(len(clips)))]
            cnt = min(hi - lo, len(src))
            first = int(rng.integers(0, len(src) - cnt + 1))
            d.append(lo); c.append(cnt); p.append(src.ctypes.data + 2 * first); keep.append(src)
        d, p, c = (np.asarray(x, np.int64) for x in (d, p, c))
        want = np.full(total, 99, np.int16)
        assert lib.ww_host_stage_i16(want.ctypes.data, total, len(d), _lib.ptr(d), _lib.ptr(p), _lib.ptr(c), 0, total, 3) == 0
        meta = rng.integers(0, 1 << 40, 0 if k == 9 else int(rng.integers(1, 4000))).astype(np.int64)
        d_pcm = torch.full((total,), 77, dtype=torch.int16, device="cuda:0")
        d_meta = torch.full((max(len(meta), 1),), -5, dtype=torch.int64, device="cuda:0")
        torch.cuda.synchronize()
        t = up.submit(total, d, p, c, d_pcm.data_ptr(), meta, d_meta.data_ptr())
        assert t == k + 1
        jobs.append((t, want, meta, d_pcm, d_meta, keep))
    # an overlapping chunk in between
    bad = up.submit(1000, np.array([0, 10], np.int64), np.array([clips[0].ctypes.data] * 2, np.int64), np.array([20, 5], np.int64),
                    jobs[0][3].data_ptr(), np.zeros(0, np.int64), 0)
    last = up.submit(10, np.zeros(0, np.int64), np.zeros(0, np.int64), np.zeros(0, np.int64), jobs[7][3].data_ptr(), np.zeros(0, np.int64), 0)
    for t, want, meta, d_pcm, d_meta, _ in jobs:
        up.wait(t, ctx)
        assert up.done(t)
        ctx.synchronize()                     # the context's stream waited for the copies on the device
        if t != 8:                            # (chunk 8's first ten samples are overwritten by `last` below)
            np.testing.assert_array_equal(d_pcm.cpu().numpy(), want)
        if len(meta):
            np.testing.assert_array_equal(d_meta.cpu().numpy(), meta)
    with pytest.raises(ValueError, match="overlap"):
        up.wait(bad, ctx)
    with pytest.raises(ValueError, match="waited for before"):
        up.wait(jobs[0][0], ctx)
    with pytest.raises(ValueError, match="no such ticket"):
        up.wait(last + 5, ctx)
    up.wait(last, ctx)
    ctx.synchronize()
    got = jobs[7][3].cpu().numpy()
    np.testing.assert_array_equal(got[:10], 0)
    np.testing.assert_array_equal(got[10:], jobs[7][1][10:])
    with pytest.raises(ValueError):
        _lib.Uploader(ctx, slots=1)
    # destroy with work still queued: the copies land before the call returns
    d_pcm = torch.zeros(200_000, dtype=torch.int16, device="cuda:0")
    src = clips[int(np.argmax([len(x) for x in clips]))]
    n = min(len(src), 150_000)
    torch.cuda.synchronize()
    up.submit(200_000, np.array([100], np.int64), np.array([src.ctypes.data], np.int64), np.array([n], np.int64), d_pcm.data_ptr(),
              np.zeros(0, np.int64), 0)
    up.close()
    torch.cuda.synchronize()
    got = d_pcm.cpu().numpy()
    np.testing.assert_array_equal(got[100:100 + n], src[:n])
    assert not got[:100].any() and not got[100 + n:].any()


def test_an_exception_inside_an_entry_point_comes_back_as_a_status():
    """no entry point throws.  ww_uploader_submit copies the caller's run arrays into std::vectors; told that there are
    2^60 runs, the copy raises std::length_error (or std::bad_alloc) inside the library - WW_GUARD_END turns it into WW_EINTERNAL /
    WW_ENOMEM, the process lives, and the uploader still serves the next, honest chunk."""
    import ctypes as C
    import torch
    from wwhip import _lib
    lib = _lib.load()
    ctx = _lib.default_context(0)
    up = _lib.Uploader(ctx, slots=2, copy_threads=2)
    try:
        one = np.zeros(1, np.int64)
        src = np.zeros(16, np.int16)
        addr = np.array([src.ctypes.data], np.int64)
        d_pcm = torch.zeros(64, dtype=torch.int16, device="cuda:0")
        ticket = C.c_int64(-1)
        rc = lib.ww_uploader_submit(up._h, 16, 1 << 60, _lib.ptr(one), _lib.ptr(addr), _lib.ptr(one), C.c_void_p(d_pcm.data_ptr()), 0, None, None, C.byref(ticket))
        assert rc in (_lib.WW_EINTERNAL, _lib.WW_ENOMEM), rc
        assert ticket.value == 0                                  # nothing was queued
        t = up.submit(16, np.array([2], np.int64), addr, np.array([5], np.int64), d_pcm.data_ptr(), np.zeros(0, np.int64), 0)
        up.wait(t, ctx)
        ctx.synchronize()
        torch.cuda.synchronize()
        assert t == 1 and not d_pcm.cpu().numpy()[:16].any()      # (the honest chunk: zeros copied where zeros were)
    finally:
        up.close()
