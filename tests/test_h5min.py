"""Minimal HDF5 reader/writer and the Keras-checkpoint cross-checks.

The fixtures under tests/golden/keras_h5 are the reference's own h5py-written Keras checkpoints
(wwdetect/CRNN/models/Arik_CRNN_data_{original,nosilence,nosilence_enhanced}/{encode,detect}.h5); the
.tflite files converted from them ship as assets/tf_lite_models/CRNN_softmax, CRNN_nosilence and
CRNN_nosilence_enhanced."""
import json
import os

import numpy as np
import pytest

from wwhip import h5min, weights


# (Keras checkpoint directory under tests/golden/keras_h5, asset directory converted from it)
KERAS_PAIRS = [("", "CRNN_softmax"), ("nosilence", "CRNN_nosilence"), ("nosilence_enhanced", "CRNN_nosilence_enhanced")]


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


@pytest.mark.parametrize("sub,asset", KERAS_PAIRS)
def test_keras_weights_equal_tflite_weights(keras_dir, assets, sub, asset):
    """Every tensor the TFLite reader assigns a role to (by graph wiring) is bit-identical to the
    tensor Keras stored under that role's name."""
    keras_dir = os.path.join(keras_dir, sub)
    c = weights.load_model_dir(os.path.join(assets, asset)).crnn
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


@pytest.mark.parametrize("sub,asset", KERAS_PAIRS)
def test_keras_semantics_agree_with_flatbuffer_oracle(keras_dir, assets, sub, asset):
    """Keras-documented layer semantics on the checkpoint vs the C restatement of the flatbuffer."""
    keras_dir = os.path.join(keras_dir, sub)
    from oracle import keras_ref
    from oracle.cpu import CpuOracle
    ora = CpuOracle(weights.pack_blob(weights.load_model_dir(os.path.join(assets, asset))))
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


@pytest.mark.parametrize("sub,asset", KERAS_PAIRS)
def test_crnn_against_pytorch_layers(keras_dir, assets, sub, asset):
    """A fifth, library-grade reading: the Keras checkpoint's weights loaded into torch.nn.Conv2d /
    torch.nn.GRU(bidirectional) / torch.nn.Linear (PyTorch's GRU is the same reset-after formulation; gates are
    ordered r, z, n instead of z, r, h) and run on the CPU, against the C restatement of the flatbuffer."""
    keras_dir = os.path.join(keras_dir, sub)
    import torch
    import torch.nn.functional as F
    from oracle.cpu import CpuOracle
    ora = CpuOracle(weights.pack_blob(weights.load_model_dir(os.path.join(assets, asset))))
    with h5min.File(os.path.join(keras_dir, "encode.h5")) as f, h5min.File(os.path.join(keras_dir, "detect.h5")) as g:
        mw, dw = f["model_weights"], g["model_weights"]
        k = torch.tensor(mw["conv2d/conv2d/kernel:0"][()]).permute(3, 2, 0, 1).contiguous()   # [out, in, kh, kw]
        kb = torch.tensor(mw["conv2d/conv2d/bias:0"][()])

        def gru_params(path):
            K, R, B = (mw[path + "/" + n][()] for n in ("kernel:0", "recurrent_kernel:0", "bias:0"))
            H = R.shape[0]
            perm = np.concatenate([np.arange(H, 2 * H), np.arange(0, H), np.arange(2 * H, 3 * H)])  # z,r,h -> r,z,n
            return (torch.tensor(K.T[perm].copy()), torch.tensor(R.T[perm].copy()), torch.tensor(B[0][perm].copy()),
                    torch.tensor(B[1][perm].copy()))

        def make_gru(inp, fwd, bwd):
            m = torch.nn.GRU(inp, 32, batch_first=True, bidirectional=True)
            with torch.no_grad():
                for sfx, p in (("", gru_params(fwd)), ("_reverse", gru_params(bwd))):
                    getattr(m, "weight_ih_l0" + sfx).copy_(p[0])
                    getattr(m, "weight_hh_l0" + sfx).copy_(p[1])
                    getattr(m, "bias_ih_l0" + sfx).copy_(p[2])
                    getattr(m, "bias_hh_l0" + sfx).copy_(p[3])
            return m.double()

        g1 = make_gru(640, "bidirectional/bidirectional/forward_gru/gru_cell_1", "bidirectional/bidirectional/backward_gru/gru_cell_2")
        g2 = make_gru(64, "bidirectional_1/bidirectional_1/forward_gru_1/gru_cell_4",
                      "bidirectional_1/bidirectional_1/backward_gru_1/gru_cell_5")
        w1, b1 = torch.tensor(dw["dense/dense/kernel:0"][()]).double(), torch.tensor(dw["dense/dense/bias:0"][()]).double()
        w2, b2 = torch.tensor(dw["dense_1/dense_1/kernel:0"][()]).double(), torch.tensor(dw["dense_1/dense_1/bias:0"][()]).double()
    rng = np.random.default_rng(9)
    wins = rng.uniform(0, 6.5, (4, 151, 40)).astype(np.float32)
    wins[1, 90:] = 0
    with torch.no_grad():
        x = torch.tensor(wins).double().permute(0, 2, 1)[:, None]                 # [B, 1, 40 (H), 151 (W)]
        # TensorFlow SAME padding for kernel (5, 20), strides (2, 8): (1, 2) on H, (6, 7) on W
        x = F.pad(x, (6, 7, 1, 2))
        y = F.relu(F.conv2d(x, k.double(), kb.double(), stride=(2, 8)))           # [B, 32, 20, 19]
        y = y.permute(0, 3, 2, 1).reshape(len(wins), 19, 640)                     # Permute((2,1,3)) + Reshape((19, 640))
        s1, _ = g1(y)
        _, h2 = g2(s1)                                                            # h2: [2, B, 32] = fwd last, bwd last
        enc = torch.cat([h2[0], h2[1]], dim=1)
        post = torch.softmax(F.relu(enc @ w1 + b1) @ w2 + b2, dim=1).numpy()
    p_o, e_o = ora.forward(wins, want_enc=True)
    assert np.abs(post - p_o).max() < 2e-6
    assert np.abs(enc.numpy() - e_o.reshape(len(wins), 64)).max() < 5e-6


def test_unsupported_hdf5_features_are_named(tmp_path):
    """Outside the 'earliest'-format subset the reader refuses by name instead of half-reading."""
    raw = bytearray(2048)
    raw[:8] = b"\x89HDF\r\n\x1a\n"
    raw[8] = 2  # superblock version 2 (libver='latest')
    raw[9] = raw[10] = 8
    p = tmp_path / "v2.h5"
    p.write_bytes(bytes(raw))
    with pytest.raises(NotImplementedError, match="superblock version 2"):
        h5min.File(str(p))
    with pytest.raises(NotImplementedError, match="compound"):
        h5min._parse_dtype(bytes([0x16, 1, 0, 0, 8, 0, 0, 0]) + bytes(24), 0)   # class 6, version 1
    with pytest.raises(NotImplementedError, match="dataspace message version 2"):
        h5min._parse_dataspace(bytes([2, 1, 0, 1]) + bytes(16), 0)


def test_h5min_reads_a_file_written_by_h5py(golden):
    """tests/golden/h5py_features.h5 was written by REAL h5py 3.3.0 / libhdf5 1.10.6 with the reference's exact calls
    (filter_dataset_to_h5.py:136-145: `create_dataset(name, data=features)` + the four attributes; generator:
    tests/golden/make_h5py_fixture.py, run in the build image).  The built-in reader must return every dataset bit for bit -
    empty, one-row and full-length ones - every attribute, the keys in h5py's name order, and load_h5 the
    evaluate_tf_lite_opts.py:35-47 arrays."""
    from wwhip.evaluate import load_h5
    z = np.load(os.path.join(golden, "h5py_features.npz"))
    names = [str(n) for n in z["names"]]
    path = os.path.join(golden, "h5py_features.h5")
    with h5min.File(path) as f:
        assert list(f.keys()) == sorted(names)
        for n in names:
            got = f[n][()]
            assert got.dtype == np.float32 and got.shape == z["f_" + n].shape
            np.testing.assert_array_equal(got, z["f_" + n])
            a = f[n].attrs
            assert [int(a[k]) for k in ("is_hotword", "speaker", "speech_start_ts", "speech_end_ts")] == z["a_" + n].tolist()
    X, y = load_h5(path, 151, 40)
    assert X.shape == (len(names), 151, 40) and y.tolist() == [int(z["a_" + n][0]) for n in sorted(names)]
    for i, n in enumerate(sorted(names)):
        rows = min(len(z["f_" + n]), 151)
        np.testing.assert_array_equal(X[i, :rows], z["f_" + n][:rows])
        assert not X[i, rows:].any()


_H5PY_READBACK = r"""
import json, sys
import h5py, numpy as np
out = {}
with h5py.File(sys.argv[1], "r") as f:
    keys = sorted(f.keys())
    for k in keys:
        d = f[k]
        out[k] = {"shape": list(d.shape), "dtype": str(d.dtype), "sum": float(np.asarray(d[()], np.float64).sum()),
                  "first": np.asarray(d[()]).ravel()[:5].tolist(),
                  "attrs": {a: int(d.attrs[a]) for a in ("is_hotword", "speaker", "speech_start_ts", "speech_end_ts")}}
np.savez(sys.argv[2], **{k: f2 for k, f2 in (("n%d" % i, h5py.File(sys.argv[1], "r")[k][()]) for i, k in enumerate(keys))})
print(json.dumps({"keys": keys, "meta": out, "h5py": h5py.__version__, "hdf5": h5py.version.hdf5_version}))
"""


def _libhdf5_python():
    """An interpreter that can import h5py: this one, or the conda one the ROCm image ships (/opt/conda: h5py 3.3.0 on
    libhdf5 1.10.6).  None if neither."""
    import shutil
    import subprocess
    import sys
    cands = [sys.executable, "/opt/conda/bin/python3.9", "/opt/conda/bin/python", shutil.which("python3.9")]
    for exe in cands:
        if exe and os.path.isfile(exe):
            try:
                if subprocess.run([exe, "-c", "import h5py, numpy"], capture_output=True, timeout=120).returncode == 0:
                    return exe
            except Exception:
                continue
    return None


def test_h5min_files_open_with_libhdf5(tmp_path):
    """A 1,000-clip feature file written by h5min - a multi-level group B-tree - read back by REAL libhdf5: every dataset and
    every attribute.  h5py is not installed for this interpreter, but the image's conda python has it (h5py 3.3.0, libhdf5
    1.10.6): the read-back runs there as a subprocess (in process where h5py imports).  filter_dataset_to_h5.py:136-145 is the
    writer this stands in for."""
    import json
    import subprocess
    exe = _libhdf5_python()
    if exe is None:
        pytest.skip("no interpreter with h5py on this machine (the build image has /opt/conda/bin/python3.9)")
    rng = np.random.default_rng(3)
    clips = {f"utt_{i:04d}_{rng.integers(1 << 20):05x}": (rng.normal(0, 1, (int(rng.integers(0, 120)), 40)).astype(np.float32),
                                                          {"is_hotword": int(i % 9 == 0), "speaker": i % 31,
                                                           "speech_start_ts": -1 if i % 4 else 7, "speech_end_ts": i % 90})
             for i in range(1000)}
    path = str(tmp_path / "feat.h5")
    h5min.write_datasets(path, clips)
    script, dump = str(tmp_path / "readback.py"), str(tmp_path / "readback.npz")
    with open(script, "w") as fh:
        fh.write(_H5PY_READBACK)
    r = subprocess.run([exe, script, path, dump], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    rep = json.loads(r.stdout.strip().splitlines()[-1])
    assert rep["keys"] == sorted(clips)
    z = np.load(dump)
    for i, name in enumerate(rep["keys"]):
        arr, attrs = clips[name]
        m = rep["meta"][name]
        assert m["shape"] == list(arr.shape) and m["dtype"] == "float32" and m["attrs"] == attrs
        np.testing.assert_array_equal(z["n%d" % i], arr)
