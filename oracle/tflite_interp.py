"""ORACLE (test infrastructure only) - op-by-op evaluator for the reference's TFLite graphs.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this package.  The product path (``wakeword-detection_amd/``)
never does.

PARITY STATUS: **unpinned for the model arithmetic**.  The reference executes
its graphs with the TensorFlow-Lite 2.4.0 interpreter (reference
``spokestack/models/tensorflow.py:26-51``, ``requirements.txt:3``); that runtime
is an un-vendored third-party dependency, absent from ``/root/reference`` and
from this image, and the reference ships no golden vectors.  This file is
therefore a restatement of the *published semantics of the TFLite builtin ops*
(tensorflow/lite/kernels reference kernels) applied to the reference's own
``.tflite`` flatbuffers, walked operator by operator exactly as wired in the
file (including WHILE subgraphs).  It shares no structure with the second
oracle (``oracle/structured.py`` / ``oracle/ww_oracle.c``), which restates the
networks from the Keras definitions (reference ``wwdetect/CRNN/model.py:21-56``,
``wwdetect/wavenet/wavenet_model.py:11-128``); the two must agree to ~1e-6
(tests/test_oracle_agreement.py).

``dtype=np.float64`` evaluates the same graph in double precision, giving a
precision-neutral value against which fp32 implementations (CPU or HIP) can be
measured.
"""
from __future__ import annotations

import os
import sys
from typing import Dict, List, Optional

import numpy as np

_PKG = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "wakeword-detection_amd")
if _PKG not in sys.path:
    sys.path.insert(0, _PKG)

from wwhip import tflite_reader as R  # noqa: E402  (parser only - no arithmetic)


class UnsupportedOp(NotImplementedError):
    pass


def _act(x: np.ndarray, code: int) -> np.ndarray:
    if code == 0:
        return x
    if code == 1:
        return np.maximum(x, 0)
    if code == 3:
        return np.clip(x, 0, 6)
    raise UnsupportedOp(f"fused activation {code}")


def _same_pads(n_in: int, k_eff: int, s: int):
    n_out = -(-n_in // s)
    total = max((n_out - 1) * s + k_eff - n_in, 0)
    return n_out, total // 2, total - total // 2


def _conv2d(x, w, b, opt, ft):
    """TFLite CONV_2D: x NHWC, w [O,KH,KW,I], explicit loops over taps (reference kernel
    order: accumulate over kh, kw, ic)."""
    n, h, wd, ci = x.shape
    co, kh, kw, ci2 = w.shape
    assert ci == ci2
    sh, sw = opt["stride_h"], opt["stride_w"]
    dh, dw = opt["dilation_h"], opt["dilation_w"]
    if opt["padding"] == "SAME":
        oh, pt, pb = _same_pads(h, (kh - 1) * dh + 1, sh)
        ow, pl, pr = _same_pads(wd, (kw - 1) * dw + 1, sw)
    else:
        oh = (h - ((kh - 1) * dh + 1)) // sh + 1
        ow = (wd - ((kw - 1) * dw + 1)) // sw + 1
        pt = pb = pl = pr = 0
    xp = np.zeros((n, h + pt + pb, wd + pl + pr, ci), dtype=ft)
    xp[:, pt : pt + h, pl : pl + wd] = x
    out = np.zeros((n, oh, ow, co), dtype=ft)
    for i in range(kh):
        for j in range(kw):
            patch = xp[:, i * dh : i * dh + (oh - 1) * sh + 1 : sh, j * dw : j * dw + (ow - 1) * sw + 1 : sw]
            out += np.einsum("nhwi,oi->nhwo", patch, w[:, i, j, :].astype(ft), dtype=ft)
    if b is not None:
        out = out + b.astype(ft)
    return _act(out, opt["activation"])


def _strided_slice(x, begin, end, strides, opt):
    if opt["ellipsis_mask"] or opt["new_axis_mask"]:
        raise UnsupportedOp("STRIDED_SLICE ellipsis/new_axis")
    idx = []
    for d in range(len(begin)):
        st = int(strides[d])
        b = None if (opt["begin_mask"] >> d) & 1 else int(begin[d])
        e = None if (opt["end_mask"] >> d) & 1 else int(end[d])
        if (opt["shrink_axis_mask"] >> d) & 1:
            bb = int(begin[d])
            if bb < 0:
                bb += x.shape[d]
            idx.append(bb)
        else:
            idx.append(slice(b, e, st))
    return x[tuple(idx)]


def _space_to_batch(x, block, pads, ft):
    # 1-D spatial form used by the Wavenet export: x [N, T, C]
    blk = int(block[0])
    p0, p1 = int(pads[0][0]), int(pads[0][1])
    n, t, c = x.shape
    xp = np.zeros((n, t + p0 + p1, c), dtype=x.dtype)
    xp[:, p0 : p0 + t] = x
    tt = xp.shape[1]
    assert tt % blk == 0
    # out[b*N + n, i] = xp[n, i*blk + b]
    y = xp.reshape(n, tt // blk, blk, c).transpose(2, 0, 1, 3).reshape(blk * n, tt // blk, c)
    return y


def _batch_to_space(x, block, crops):
    blk = int(block[0])
    c0, c1 = int(crops[0][0]), int(crops[0][1])
    bn, t, c = x.shape
    n = bn // blk
    y = x.reshape(blk, n, t, c).transpose(1, 2, 0, 3).reshape(n, t * blk, c)
    return y[:, c0 : t * blk - c1]


class Interpreter:
    """Evaluate one ``.tflite`` model.  ``__call__(*inputs) -> list of outputs``."""

    def __init__(self, path_or_model, dtype=np.float32) -> None:
        self.model = R.load(path_or_model) if isinstance(path_or_model, str) else path_or_model
        self.ft = dtype

    # -- public -----------------------------------------------------------
    def __call__(self, *inputs) -> List[np.ndarray]:
        return self._run(0, list(inputs))

    # -- internals ----------------------------------------------------------
    def _c(self, a: np.ndarray) -> np.ndarray:
        if a.dtype in (np.float32, np.float64, np.float16):
            return a.astype(self.ft)
        return a

    def _run(self, sgi: int, inputs: List[np.ndarray]) -> List[np.ndarray]:
        sg = self.model.subgraphs[sgi]
        val: Dict[int, np.ndarray] = {}
        for t in sg.tensors:
            if t.data is not None:
                val[t.index] = self._c(t.data)
        assert len(inputs) == len(sg.inputs), (len(inputs), len(sg.inputs))
        for i, a in zip(sg.inputs, inputs):
            val[i] = self._c(np.asarray(a))
        for op in sg.operators:
            outs = self._eval(sg, op, [val[i] if i >= 0 else None for i in op.inputs])
            if not isinstance(outs, (list, tuple)):
                outs = [outs]
            for i, o in zip(op.outputs, outs):
                val[i] = o
        return [val[i] for i in sg.outputs]

    def _eval(self, sg, op, x):
        ft = self.ft
        o = op.options
        name = op.op
        if name == "CONV_2D":
            return _conv2d(x[0], x[1], x[2] if len(x) > 2 else None, o, ft)
        if name == "FULLY_CONNECTED":
            y = x[0].reshape(-1, x[1].shape[1]).astype(ft) @ x[1].astype(ft).T
            if len(x) > 2 and x[2] is not None:
                y = y + x[2]
            return _act(y.astype(ft), o.get("activation", 0))
        if name == "ADD":
            return _act((x[0] + x[1]), o.get("activation", 0))
        if name == "SUB":
            return _act((x[0] - x[1]), o.get("activation", 0))
        if name == "MUL":
            return _act((x[0] * x[1]), o.get("activation", 0))
        if name == "MAXIMUM":
            return np.maximum(x[0], x[1])
        if name == "LOG":
            return np.log(x[0])
        if name == "LOGISTIC":
            return (1.0 / (1.0 + np.exp(-x[0].astype(ft)))).astype(ft)
        if name == "TANH":
            return np.tanh(x[0].astype(ft)).astype(ft)
        if name == "RELU":
            return np.maximum(x[0], 0)
        if name == "RESHAPE":
            return x[0].reshape(tuple(int(v) for v in x[1]))
        if name == "TRANSPOSE":
            return np.transpose(x[0], tuple(int(v) for v in x[1]))
        if name == "SHAPE":
            return np.array(x[0].shape, dtype=np.int32)
        if name == "STRIDED_SLICE":
            return _strided_slice(x[0], x[1], x[2], x[3], o)
        if name == "PACK":
            return np.stack(x, axis=o["axis"])
        if name == "FILL":
            return np.full(tuple(int(v) for v in x[0]), x[1], dtype=x[1].dtype)
        if name == "REVERSE_V2":
            return np.flip(x[0], axis=tuple(int(v) for v in np.atleast_1d(x[1])))
        if name == "CONCATENATION":
            return _act(np.concatenate(x, axis=o["axis"]), o.get("activation", 0))
        if name == "SPLIT":
            return list(np.split(x[1], o["num_splits"], axis=int(x[0])))
        if name == "GATHER":
            return np.take(x[0], x[1], axis=o["axis"])
        if name == "SLICE":
            begin = [int(v) for v in x[1]]
            size = [int(v) for v in x[2]]
            idx = tuple(slice(b, None if s == -1 else b + s) for b, s in zip(begin, size))
            return x[0][idx]
        if name == "EXPAND_DIMS":
            return np.expand_dims(x[0], int(x[1]))
        if name == "CAST":
            out_t = sg.tensors[op.outputs[0]].dtype
            return x[0] if np.issubdtype(out_t, np.floating) else x[0].astype(out_t)
        if name == "LESS":
            return np.less(x[0], x[1])
        if name == "PAD":
            return np.pad(x[0], [(int(a), int(b)) for a, b in x[1]])
        if name == "SPACE_TO_BATCH_ND":
            return _space_to_batch(x[0], x[1], x[2], ft)
        if name == "BATCH_TO_SPACE_ND":
            return _batch_to_space(x[0], x[1], x[2])
        if name == "REDUCE_MAX":
            ax = tuple(int(v) for v in np.atleast_1d(x[1]))
            return np.max(x[0], axis=ax, keepdims=o.get("keep_dims", False))
        if name == "SOFTMAX":
            z = x[0].astype(ft) * ft(o.get("beta", 1.0))
            z = z - np.max(z, axis=-1, keepdims=True)
            e = np.exp(z)
            return (e / np.sum(e, axis=-1, keepdims=True)).astype(ft)
        if name == "WHILE":
            state = list(x)
            guard = 0
            while bool(self._run(o["cond_subgraph"], state)[0]):
                state = self._run(o["body_subgraph"], state)
                guard += 1
                if guard > 100000:
                    raise RuntimeError("WHILE did not terminate")
            return state
        raise UnsupportedOp(name)


class ModelDir:
    """filter/encode/detect triple evaluated the way the reference calls them
    (``utils/evaluate_models.py:76-86``)."""

    def __init__(self, model_dir: str, dtype=np.float32) -> None:
        self.filter = Interpreter(os.path.join(model_dir, "filter.tflite"), dtype)
        self.encode = Interpreter(os.path.join(model_dir, "encode.tflite"), dtype)
        self.detect = Interpreter(os.path.join(model_dir, "detect.tflite"), dtype)
        self.is_crnn = any(op.op == "WHILE" for op in self.encode.model.main.operators)

    def mel(self, magnitude: np.ndarray) -> np.ndarray:
        """[257] -> [40]  (reference utils/tf_lite/filter.py:70-75)."""
        return self.filter(np.asarray(magnitude)[None, :])[0][0]

    def window(self, mel_window: np.ndarray) -> np.ndarray:
        """[T,40] -> detect output row (reference utils/evaluate_models.py:76-86)."""
        if self.is_crnn:
            x = np.asarray(mel_window).T[None, :, :, None]
        else:
            x = np.asarray(mel_window)[None]
        enc = self.encode(x)[0]
        return self.detect(enc)[0][0]
