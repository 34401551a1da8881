"""The two CPU oracles against each other, the committed fixtures and SURVEY Appendix B."""
import os

import numpy as np
import pytest

from oracle import cpu, numpy_ref as NR
from oracle.tflite_interp import ModelDir
from wwhip import weights as W

MODELS = ["CRNN", "CRNN_softmax", "Wavenet", "Wavenet_alt", "CRNN_nosilence", "CRNN_nosilence_enhanced", "CRNN_old"]


@pytest.fixture(scope="module")
def oracles(assets):
    return {m: cpu.CpuOracle(W.pack_blob(W.load_model_dir(os.path.join(assets, m)))) for m in MODELS}


def test_appendix_b_values(assets):
    m = ModelDir(os.path.join(assets, "CRNN"))
    assert m.window(np.zeros((151, 40), np.float32))[0] == pytest.approx(0.00343328, abs=2e-8)
    assert m.window(np.full((151, 40), 3.0, np.float32))[0] == pytest.approx(0.15505268, abs=2e-7)
    w = ModelDir(os.path.join(assets, "Wavenet"))
    np.testing.assert_allclose(w.window(np.zeros((182, 40), np.float32)), [0.87389916, 0.12610082], atol=2e-7)
    n = np.arange(512)
    mag = np.abs(np.fft.rfft(0.5 * np.sin(2 * np.pi * 1000 * n / 16000) * np.hanning(512), n=512)).astype(np.float32)
    mel = m.mel(mag)
    assert mel.argmax() == 13
    np.testing.assert_allclose(mel[12:16], [5.5934095, 5.7318573, 1.9074972, 1.1997461], atol=2e-6)
    assert np.all(m.mel(np.zeros(257, np.float32)) == 0.0)


@pytest.mark.parametrize("name", MODELS)
def test_c_restatement_matches_op_by_op_fixtures(oracles, golden, name):
    z = np.load(os.path.join(golden, "models.npz"))
    out, enc = oracles[name].forward(z[name + ".windows"], want_enc=True)
    assert np.abs(out - z[name + ".det32"]).max() < 1e-6
    assert np.abs(out - z[name + ".det64"]).max() < 1e-6
    assert np.abs(enc.reshape(z[name + ".enc32"].shape) - z[name + ".enc32"]).max() < 2e-5


def test_c_front_end_matches_reference_numpy_fixtures(oracles, golden):
    z = np.load(os.path.join(golden, "frontend.npz"))
    o = oracles["CRNN"]
    for n in ["noise_chirp", "quiet", "silence", "fullscale", "ragged"]:
        got = o.logmel(z[n + ".pcm"], divisor=32767.0, clip=True, preemph=0.0)
        assert got.shape == z[n + ".div32767.mel"].shape
        assert np.abs(got - z[n + ".div32767.mel"]).max() < 2e-5
        got = o.logmel(z[n + ".pcm"], divisor=32768.0, clip=False, preemph=0.97)
        assert np.abs(got - z[n + ".div32768.mel"]).max() < 2e-5
        mag = cpu.stft_mag(np.lib.stride_tricks.sliding_window_view(
            np.clip(z[n + ".pcm"].astype(np.float32) / np.float32(32767), -1, 1), 512)[::160])
        np.testing.assert_allclose(mag, z[n + ".div32767.mag"], rtol=3e-6, atol=1e-9)


def test_ref_filter_matches_c_oracle_streaming(oracles, assets):
    m = ModelDir(os.path.join(assets, "CRNN"))
    rng = np.random.default_rng(4)
    x = rng.normal(0, 0.1, 3200).astype(np.float32)
    f = NR.RefFilter(lambda a: m.filter(a)[0], pre_emphasis=0.5)
    feats = []
    for s in range(0, 3200, 320):
        feats += f.filter_frame(x[s:s + 320].copy())
    want = oracles["CRNN"].logmel_f32(x, preemph=0.5)
    assert len(feats) == 17 and np.abs(np.array(feats) - want).max() < 2e-5


@pytest.mark.parametrize("case", ["short", "long", "exact30"])
def test_evaluator_oracle_vs_numpy_fixture(golden, case):
    z = np.load(os.path.join(golden, "evaluator.npz"))
    sm = cpu.smooth(z[case + ".neg"], 30)
    # np.convolve sums the 30 products in BLAS order (implementation detail): agreement to 1-2 ulp
    np.testing.assert_allclose(sm, z[case + ".smoothed"], rtol=0, atol=1e-15)
    frr, fa, cnt = cpu.far_frr(z[case + ".pos"], z[case + ".smoothed"], np.arange(0.5, 0.99999, 0.005), 200,
                               float(z[case + ".hours"][0]))
    np.testing.assert_array_equal(cnt, z[case + ".cnt"])
    np.testing.assert_allclose(frr, z[case + ".frr"], atol=1e-15)


def test_frr_at_fa_metric():
    frr = np.array([0.0, 0.1, 0.2, 0.5])
    far = np.array([3.0, 0.6, 0.5, 0.0])
    assert NR.frr_at_fa(frr, far, 0.5) == 0.2
    assert np.isnan(NR.frr_at_fa(frr, far + 10, 0.5))


def test_wfst_smooth_restatement_is_the_shortest_path():
    """oracle/numpy_ref.wfst_smooth (wwdetect/wfst.py:17-71 without pynini) against brute force over all
    2^T state sequences, and on the two superframes the reference's __main__ exercises."""
    import itertools
    from oracle import numpy_ref as NR

    def brute(pp):
        obs = -np.log(np.asarray(pp, np.float64))
        best = None
        for path in itertools.product((0, 1), repeat=len(obs)):
            c = np.log(2) + obs[0, path[0]]
            for t in range(1, len(obs)):
                c += obs[t, path[t]] - (1 if path[t] == path[t - 1] else 0)
            if best is None or c < best[0] - 1e-12:
                best = (c, path)
        return list(best[1])

    rng = np.random.default_rng(0)
    for _ in range(120):
        T = int(rng.integers(1, 11))
        p = rng.uniform(0.02, 0.98, T)
        pp = np.stack([1 - p, p], 1)
        assert NR.wfst_smooth(pp) == brute(pp)
    t1 = [[0.8, 0.2], [0.9, 0.1], [0.5, 0.5], [0.4, 0.6], [0.2, 0.8], [0.6, 0.4], [0.3, 0.7], [0.4, 0.6], [0.5, 0.5], [0.9, 0.1]]
    t2 = [[0.8, 0.2], [0.9, 0.1], [0.5, 0.5], [0.55, 0.45], [0.2, 0.8], [0.6, 0.4], [0.7, 0.3], [0.8, 0.2], [0.3, 0.7], [0.9, 0.1]]
    assert NR.wfst_smooth(t1) == [0, 0, 0, 1, 1, 1, 1, 1, 0, 0]   # stays in 'wakeword' through the 0.6/0.4 dip (wfst.py:83)
    assert NR.wfst_smooth(t2) == [0] * 10                          # does not enter on one errant frame (wfst.py:94)


@pytest.mark.parametrize("name", ["Wavenet", "Wavenet_alt"])
def test_wavenet_against_pytorch_layers(assets, name):
    """wwdetect/wavenet/wavenet_model.py:11-128 rebuilt from torch.nn.functional (conv1d with dilation on a
    causally padded input, the inference-mode BatchNorm affine, GlobalMaxPooling1D + softmax), fed with the
    weights the TFLite reader extracts, float64 - PyTorch's convolution kernels instead of the flatbuffer's
    PAD / SPACE_TO_BATCH / CONV_2D / BATCH_TO_SPACE chains that the op-by-op interpreter walks."""
    import torch
    import torch.nn.functional as F
    from wwhip import weights
    from oracle.cpu import CpuOracle
    bundle = weights.load_model_dir(os.path.join(assets, name))
    w = bundle.wavenet
    ora = CpuOracle(weights.pack_blob(bundle))
    T = lambda a: torch.tensor(np.asarray(a)).double()

    def conv1x1(x, wt, b, act=None):          # x [B, C, T]; wt [C_in, C_out]
        y = F.conv1d(x, T(wt).t()[:, :, None], T(b))
        return F.relu(y) if act == "relu" else y

    def causal(x, wt3, b, d):                 # wt3 [3, C_in, C_out], tap k reads x[t - (2 - k) d]
        k = T(wt3).permute(2, 1, 0).contiguous()   # [C_out, C_in, 3]
        return F.conv1d(F.pad(x, (2 * d, 0)), k, T(b), dilation=d)

    rng = np.random.default_rng(19)
    wins = rng.uniform(0, 6.5, (3, w.n_frames, w.n_mel)).astype(np.float32)
    wins[1, 120:] = 0
    with torch.no_grad():
        x = conv1x1(T(wins).permute(0, 2, 1), w.w_in, w.b_in, "relu")
        skips = []
        for b in w.blocks:
            u = x * T(b.bn_scale)[None, :, None] + T(b.bn_shift)[None, :, None]
            merged = torch.tanh(causal(u, b.w_tanh, b.b_tanh, b.dilation)) * torch.sigmoid(causal(u, b.w_sig, b.b_sig, b.dilation))
            skips.append(conv1x1(merged, b.w_skip, b.b_skip, "relu"))
            if b.w_res is not None:
                x = conv1x1(merged, b.w_res, b.b_res, "relu") + x
        enc = sum(skips[i] for i in w.skip_order)
        h = conv1x1(F.relu(enc), w.det_w1, w.det_b1, "relu")
        y = conv1x1(h, w.det_w2, w.det_b2).max(dim=2).values
        post = torch.softmax(y, dim=1).numpy()
    p_o, e_o = ora.forward(wins, want_enc=True)
    assert np.abs(post - p_o).max() < 2e-6
    assert np.abs(enc.permute(0, 2, 1).numpy() - e_o).max() < 2e-5 * max(1.0, float(np.abs(e_o).max()))
