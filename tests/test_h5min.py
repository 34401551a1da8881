"""Minimal HDF5 reader/writer and the Keras-checkpoint cross-checks.

The fixtures under tests/golden/keras_h5 are the reference's own h5py-written Keras checkpoints
(wwdetect/CRNN/models/Arik_CRNN_data_original/{encode,detect}.h5); the .tflite files converted
from them ship as assets/tf_lite_models/CRNN_softmax."""
import json
import os

import numpy as np
import pytest

from wwhip import h5min, weights


@pytest.fixture(scope="module")
def keras_dir(golden):
    return os.path.join(golden, "keras_h5")


def test_reads_h5py_written_checkpoint(keras_dir):
    with h5min.File(os.path.join(keras_dir, "encode.h5")) as f:
        assert f.keys() == ["model_weights"]
        assert f.attrs["keras_version"] == "2.4.0" and f.attrs["backend"] == "tensorflow"  # vlen UTF-8 strings
        cfg = json.loads(f.attrs["model_config"])
        assert [L["class_name"] for L in cfg["config"]["layers"]] == \
            ["InputLayer", "Conv2D", "Permute", "Reshape", "Bidirectional", "Bidirectional"]
        mw = f["model_weights"]
        assert list(mw.attrs["layer_names"]) == [b"conv2d", b"permute", b"reshape", b"bidirectional", b"bidirectional_1"]
        assert list(mw["conv2d"].attrs["weight_names"]) == [b"conv2d/kernel:0", b"conv2d/bias:0"]
        assert len(mw["permute"].attrs["weight_names"]) == 0
        k = mw["conv2d/conv2d/kernel:0"]
        assert k.shape == (5, 20, 1, 32) and k.dtype == np.float32
        assert mw["bidirectional"]["bidirectional/forward_gru/gru_cell_1/kernel:0"].shape == (640, 96)
        assert "conv2d" in mw and "nope" not in mw
        with pytest.raises(KeyError):
            mw["conv2d/nope"]


def test_keras_weights_equal_tflite_weights(keras_dir, assets):
    """Every tensor the TFLite reader assigns a role to (by graph wiring) is bit-identical to the
    tensor Keras stored under that role's name."""
    c = weights.load_model_dir(os.path.join(assets, "CRNN_softmax")).crnn
    with h5min.File(os.path.join(keras_dir, "encode.h5")) as f, h5min.File(os.path.join(keras_dir, "detect.h5")) as g:
        mw, dw = f["model_weights"], g["model_weights"]
        np.testing.assert_array_equal(np.transpose(mw["conv2d/conv2d/kernel:0"][()][:, :, 0, :], (2, 0, 1)), c.conv_w)
        np.testing.assert_array_equal(mw["conv2d/conv2d/bias:0"][()], c.conv_b)
        for path, gd in [("bidirectional/bidirectional/forward_gru/gru_cell_1", c.gru1[0]),
                         ("bidirectional/bidirectional/backward_gru/gru_cell_2", c.gru1[1]),
                         ("bidirectional_1/bidirectional_1/forward_gru_1/gru_cell_4", c.gru2[0]),
                         ("bidirectional_1/bidirectional_1/backward_gru_1/gru_cell_5", c.gru2[1])]:
            np.testing.assert_array_equal(mw[path + "/kernel:0"][()].T, gd.w_x)
            np.testing.assert_array_equal(mw[path + "/recurrent_kernel:0"][()].T, gd.w_h)
            np.testing.assert_array_equal(mw[path + "/bias:0"][()][0], gd.b_x)
            np.testing.assert_array_equal(mw[path + "/bias:0"][()][1], gd.b_h)
        np.testing.assert_array_equal(dw["dense/dense/kernel:0"][()].T, c.head_w1)
        np.testing.assert_array_equal(dw["dense/dense/bias:0"][()], c.head_b1)
        np.testing.assert_array_equal(dw["dense_1/dense_1/kernel:0"][()].T, c.head_w2)
        np.testing.assert_array_equal(dw["dense_1/dense_1/bias:0"][()], c.head_b2)
        cfg = json.loads(f.attrs["model_config"])["config"]["layers"]
    conv = cfg[1]["config"]
    assert conv["strides"] == [c.stride_f, c.stride_t] and conv["padding"] == "same" and conv["activation"] == "relu"
    gru = cfg[4]["config"]["layer"]["config"]
    assert gru["reset_after"] and gru["recurrent_activation"] == "sigmoid" and gru["activation"] == "tanh"
    assert gru["units"] == c.units and cfg[4]["config"]["merge_mode"] == "concat"


def test_keras_semantics_agree_with_flatbuffer_oracle(keras_dir, assets):
    """Keras-documented layer semantics on the checkpoint vs the C restatement of the flatbuffer."""
    from oracle import keras_ref
    from oracle.cpu import CpuOracle
    ora = CpuOracle(weights.pack_blob(weights.load_model_dir(os.path.join(assets, "CRNN_softmax"))))
    rng = np.random.default_rng(5)
    w = rng.uniform(0, 6.5, (5, 151, 40)).astype(np.float32)
    w[1] = 0
    w[2, 100:] = 0
    p_k, e_k = keras_ref.crnn_forward(os.path.join(keras_dir, "encode.h5"), os.path.join(keras_dir, "detect.h5"), w)
    p_o, e_o = ora.forward(w, want_enc=True)
    assert np.abs(p_k - p_o).max() < 2e-6
    assert np.abs(e_k - e_o.reshape(e_k.shape)).max() < 5e-6


def test_feature_file_roundtrip(tmp_path):
    """The reference's feature-file layout (filter_dataset_to_h5.py:136-145), enough clips for a
    two-level group B-tree (> 32 symbol nodes of 8 entries)."""
    rng = np.random.default_rng(0)
    clips = {}
    for i in range(300):
        T = int(rng.integers(0, 200)) if i else 0  # one empty clip
        clips[f"clip_{rng.integers(1 << 30):08x}_{i}"] = (
            rng.normal(0, 1, (T, 40)).astype(np.float32),
            {"is_hotword": int(i % 7 == 0), "speaker": int(i % 13), "speech_start_ts": -1 if i % 5 else 12,
             "speech_end_ts": float(i) / 3, "note": f"clip {i} é", "flag": bool(i & 1),
             "vec": np.arange(3, dtype=np.int16)})
    path = str(tmp_path / "test.h5")
    h5min.write_datasets(path, clips, root_attrs={"made_by": "h5min"})
    with h5min.File(path) as f:
        assert f.keys() == sorted(clips)
        assert f.attrs["made_by"] == "h5min"
        for name, (arr, attrs) in clips.items():
            d = f[name]
            assert d.shape == arr.shape and d.dtype == np.float32
            np.testing.assert_array_equal(d[()], arr)
            a = d.attrs
            assert a["is_hotword"] == attrs["is_hotword"] and a["speaker"] == attrs["speaker"]
            assert a["speech_start_ts"] == attrs["speech_start_ts"] and a["speech_end_ts"] == attrs["speech_end_ts"]
            assert a["note"] == attrs["note"] and bool(a["flag"]) is attrs["flag"]
            np.testing.assert_array_equal(a["vec"], attrs["vec"])
    # header sanity a foreign reader relies on: signature, version-0 superblock, end-of-file address
    raw = open(path, "rb").read()
    assert raw[:8] == b"\x89HDF\r\n\x1a\n" and raw[8] == 0
    assert int.from_bytes(raw[40:48], "little") == len(raw)


def test_load_h5_matches_reference_loader_semantics(tmp_path):
    from wwhip.evaluate import load_h5
    rng = np.random.default_rng(1)
    feats = {"b": rng.normal(0, 1, (200, 40)).astype(np.float32), "a": rng.normal(0, 1, (90, 40)).astype(np.float32),
             "c": np.zeros((0, 40), np.float32)}
    path = str(tmp_path / "t.h5")
    h5min.write_datasets(path, {k: (v, {"is_hotword": int(k == "a")}) for k, v in feats.items()})
    X, y = load_h5(path, 151, 40)
    assert X.shape == (3, 151, 40) and y.tolist() == [1, 0, 0]  # keys in name order: a, b, c
    np.testing.assert_array_equal(X[0, :90], feats["a"])
    assert not X[0, 90:].any()
    np.testing.assert_array_equal(X[1], feats["b"][:151])
    assert not X[2].any()


def test_not_hdf5(tmp_path):
    p = tmp_path / "x.h5"
    p.write_bytes(b"not an hdf5 file at all" * 10)
    with pytest.raises(ValueError):
        h5min.File(str(p))
