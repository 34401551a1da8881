"""Flatbuffer reader + weight extraction by wiring (host logic, CPU only)."""
import os

import numpy as np
import pytest

from wwhip import tflite_reader as R
from wwhip import weights as W

MODELS = ["CRNN", "CRNN_softmax", "Wavenet", "Wavenet_alt", "CRNN_nosilence", "CRNN_nosilence_enhanced", "CRNN_old"]


def test_filter_graph_constants(assets):
    f = W.extract_filter(R.load(os.path.join(assets, "CRNN", "filter.tflite")))
    assert f.weight.shape == (40, 257)
    assert int((f.weight != 0).sum()) == 490
    assert np.all(f.bias == 0)
    assert f.floor == pytest.approx(1e-5, rel=1e-6)
    assert f.log_offset == pytest.approx(11.512925, rel=1e-7)
    assert f.scale == 0.5


def test_crnn_geometry_and_param_count(assets):
    b = W.load_model_dir(os.path.join(assets, "CRNN"))
    c = b.crnn
    assert (c.n_mel, c.n_frames) == (40, 151)
    assert c.conv_w.shape == (32, 5, 20) and (c.stride_f, c.stride_t) == (2, 8)
    assert c.pad_f == (1, 2) and c.pad_t == (6, 7) and (c.out_f, c.out_t) == (20, 19)
    n = c.conv_w.size + c.conv_b.size
    for g in (*c.gru1, *c.gru2):
        n += g.w_x.size + g.b_x.size + g.w_h.size + g.b_h.size
    n += c.head_w1.size + c.head_b1.size + c.head_w2.size + c.head_b2.size
    assert n == 155681
    assert c.head_kind == W.HEAD_SIGMOID and b.posterior_index == 0
    assert c.gru1[0].w_x.shape == (96, 640) and c.gru2[1].w_x.shape == (96, 64)


def test_softmax_head_variant(assets):
    b = W.load_model_dir(os.path.join(assets, "CRNN_softmax"))
    assert b.crnn.head_kind == W.HEAD_SOFTMAX and b.n_out == 2 and b.posterior_index == 1


def test_wavenet_structure(assets):
    b = W.load_model_dir(os.path.join(assets, "Wavenet"))
    w = b.wavenet
    assert (w.n_frames, w.n_mel, w.channels, w.skip_channels) == (182, 40, 16, 32)
    assert [blk.dilation for blk in w.blocks] == [1, 2, 4, 8] * 6
    assert [blk.w_res is None for blk in w.blocks] == [False] * 23 + [True]   # last residual pruned
    assert w.skip_order == list(range(24))
    assert 1 + 6 * 2 * (1 + 2 + 4 + 8) == 181                                  # receptive field


@pytest.mark.parametrize("name", MODELS)
def test_blob_roundtrip(assets, name):
    b = W.load_model_dir(os.path.join(assets, name))
    blob = W.pack_blob(b)
    u = W.unpack_blob(blob)
    assert int(u["__kind__"][0]) == b.kind
    assert np.array_equal(u["filter.w"].reshape(40, 257), b.filt.weight)
    if b.kind == W.KIND_CRNN:
        assert np.array_equal(u["crnn.g1b.wx"].reshape(96, -1), b.crnn.gru1[1].w_x)
    else:
        assert np.array_equal(u["wave.w_tanh"].reshape(24, 3, 16, 16)[5], b.wavenet.blocks[5].w_tanh)
    assert len(blob) % 16 == 0


def test_io_details_shapes(assets):
    i, o = R.io_details(R.load(os.path.join(assets, "Wavenet", "encode.tflite")))
    assert list(i[0]["shape"]) == [1, 182, 40] and list(o[0]["shape"]) == [1, 182, 32]
    i, _ = R.io_details(R.load(os.path.join(assets, "CRNN", "encode.tflite")))
    assert list(i[0]["shape"]) == [1, 40, 151, 1]


def test_reader_rejects_garbage():
    with pytest.raises(R.FlatBufferError):
        R.parse(b"\x00" * 64)


def test_missing_model_file(tmp_path):
    with pytest.raises(FileNotFoundError):
        W.load_model_dir(str(tmp_path))


def test_filter_weights_are_the_slaney_mel_filterbank(assets):
    """The 40 x 257 matrix read out of filter.tflite is librosa.filters.mel(sr=16000, n_fft=512, n_mels=40,
    fmin=0, fmax=8000, htk=False, norm='slaney') - restated here from its published formula - to fp32
    rounding: pins the extraction (orientation, band order) against something the reader did not produce."""
    from wwhip import weights

    def hz2mel(f):
        f = np.asarray(f, float)
        f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, np.log(6.4) / 27.0
        return np.where(f >= min_log_hz, min_log_hz / f_sp + np.log(np.maximum(f, 1e-9) / min_log_hz) / logstep, f / f_sp)

    def mel2hz(m):
        m = np.asarray(m, float)
        f_sp, min_log_hz, logstep = 200.0 / 3, 1000.0, np.log(6.4) / 27.0
        return np.where(m >= min_log_hz / f_sp, min_log_hz * np.exp(logstep * (m - min_log_hz / f_sp)), f_sp * m)

    fft = np.linspace(0, 8000.0, 257)
    mel_f = mel2hz(np.linspace(hz2mel(0.0), hz2mel(8000.0), 42))
    fdiff, ramps = np.diff(mel_f), mel_f[:, None] - fft[None]
    M = np.stack([np.maximum(0, np.minimum(-ramps[i] / fdiff[i], ramps[i + 2] / fdiff[i + 1])) for i in range(40)])
    M *= (2.0 / (mel_f[2:42] - mel_f[:40]))[:, None]
    for name in ("CRNN", "Wavenet"):
        f = weights.load_model_dir(os.path.join(assets, name)).filt
        assert f.weight.shape == (40, 257)
        assert np.abs(f.weight - M).max() < 2e-9
        assert not f.bias.any()
        # y = 0.5 * (ln(max(x, 1e-5)) - ln(1e-5))
        assert abs(f.floor - 1e-5) < 1e-12 and abs(f.log_offset + np.log(1e-5)) < 1e-6 and f.scale == 0.5


def test_old_crnn_export_geometry(assets):
    """utils/CRNN_files/{encode,detect}_old.tflite: same op set, different conv (20x5 kernel over freq x time,
    stride 8x2, VALID) -> [74 steps][3*32 features]; sigmoid head."""
    b = W.load_model_dir(os.path.join(assets, "CRNN_old"))
    c = b.crnn
    assert c.conv_w.shape == (32, 20, 5) and (c.stride_f, c.stride_t) == (8, 2)
    assert c.pad_f == (0, 0) and c.pad_t == (0, 0) and (c.out_f, c.out_t) == (3, 74)
    assert c.gru1[0].w_x.shape == (96, 96) and c.gru2[0].w_x.shape == (96, 64)
    assert c.head_kind == W.HEAD_SIGMOID and b.posterior_index == 0


@pytest.mark.parametrize("name,probs", [("CRNN_softmax", [0.8681737, 0.13182628]),
                                        ("CRNN_nosilence_enhanced", [0.5368042, 0.46319583])])
def test_softmax_variants_on_zeros_match_survey_appendix_b(assets, name, probs):
    from oracle.tflite_interp import ModelDir
    got = ModelDir(os.path.join(assets, name)).window(np.zeros((151, 40), np.float32))
    np.testing.assert_allclose(got, probs, atol=2e-7)
