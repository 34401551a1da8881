"""Minimal reader for TFLite (schema v3, "TFL3") flatbuffers.

The reference executes ``filter.tflite`` / ``encode.tflite`` / ``detect.tflite``
through the TensorFlow-Lite interpreter (reference
``spokestack/models/tensorflow.py:24-51``).  This build never runs TFLite: it
only needs the *contents* of those files (graph wiring, tensor shapes and the
fp32 weight buffers), which this module decodes straight from the bytes.

Only the flatbuffer primitives used by the TFLite schema are implemented:
root table, vtable field lookup, scalars, strings, vectors of scalars and
vectors of tables.  Field ids follow ``schema_v3.fbs`` of TensorFlow 2.4
(the version pinned by the reference, ``requirements.txt:3``).
"""
from __future__ import annotations

import struct
from dataclasses import dataclass, field
from typing import Dict, List, Optional

import numpy as np

# builtin operator codes that occur in the shipped graphs
BUILTIN = {
    0: "ADD", 2: "CONCATENATION", 3: "CONV_2D", 9: "FULLY_CONNECTED",
    14: "LOGISTIC", 18: "MUL", 19: "RELU", 22: "RESHAPE", 25: "SOFTMAX",
    28: "TANH", 34: "PAD", 36: "GATHER", 37: "BATCH_TO_SPACE_ND",
    38: "SPACE_TO_BATCH_ND", 39: "TRANSPOSE", 41: "SUB", 45: "STRIDED_SLICE",
    49: "SPLIT", 53: "CAST", 55: "MAXIMUM", 58: "LESS", 65: "SLICE",
    70: "EXPAND_DIMS", 73: "LOG", 77: "SHAPE", 82: "REDUCE_MAX", 83: "PACK",
    94: "FILL", 105: "REVERSE_V2", 119: "WHILE", 40: "MEAN", 42: "DIV",
    43: "SQUEEZE", 88: "UNPACK", 61: "GREATER", 86: "LOGICAL_AND",
    74: "SUM", 57: "ARG_MAX", 1: "AVERAGE_POOL_2D", 17: "MAX_POOL_2D",
    47: "EXP", 54: "PRELU", 76: "RSQRT", 75: "SQRT", 92: "SQUARE",
    99: "SQUARED_DIFFERENCE", 116: "DENSIFY", 6: "DEQUANTIZE", 114: "QUANTIZE",
    130: "BROADCAST_TO", 100: "MIRROR_PAD", 60: "PADV2", 102: "SPLIT_V",
    101: "ABS", 59: "NEG", 63: "SELECT", 123: "SELECT_V2", 127: "BATCH_MATMUL",
}

TENSOR_DTYPE = {
    0: np.float32, 1: np.float16, 2: np.int32, 3: np.uint8, 4: np.int64,
    6: np.bool_, 7: np.int16, 9: np.int8, 10: np.float64,
}


class FlatBufferError(ValueError):
    """Raised when the bytes are not a TFLite v3 flatbuffer we understand."""


class _Table:
    """A flatbuffer table: position + vtable view over an immutable buffer."""

    __slots__ = ("buf", "pos", "vt", "vt_len")

    def __init__(self, buf: bytes, pos: int) -> None:
        self.buf = buf
        self.pos = pos
        self.vt = pos - struct.unpack_from("<i", buf, pos)[0]
        self.vt_len = struct.unpack_from("<H", buf, self.vt)[0]

    def _off(self, fid: int) -> int:
        slot = 4 + 2 * fid
        if slot >= self.vt_len:
            return 0
        return struct.unpack_from("<H", self.buf, self.vt + slot)[0]

    def scalar(self, fid: int, fmt: str, default=0):
        o = self._off(fid)
        if not o:
            return default
        return struct.unpack_from("<" + fmt, self.buf, self.pos + o)[0]

    def _indirect(self, fid: int) -> Optional[int]:
        o = self._off(fid)
        if not o:
            return None
        p = self.pos + o
        return p + struct.unpack_from("<I", self.buf, p)[0]

    def table(self, fid: int) -> Optional["_Table"]:
        p = self._indirect(fid)
        return None if p is None else _Table(self.buf, p)

    def string(self, fid: int) -> str:
        p = self._indirect(fid)
        if p is None:
            return ""
        n = struct.unpack_from("<I", self.buf, p)[0]
        return self.buf[p + 4 : p + 4 + n].decode("utf-8", "replace")

    def vector(self, fid: int, dtype) -> np.ndarray:
        p = self._indirect(fid)
        if p is None:
            return np.zeros(0, dtype=dtype)
        n = struct.unpack_from("<I", self.buf, p)[0]
        return np.frombuffer(self.buf, dtype=dtype, count=n, offset=p + 4)

    def tables(self, fid: int) -> List["_Table"]:
        p = self._indirect(fid)
        if p is None:
            return []
        n = struct.unpack_from("<I", self.buf, p)[0]
        out = []
        for i in range(n):
            e = p + 4 + 4 * i
            out.append(_Table(self.buf, e + struct.unpack_from("<I", self.buf, e)[0]))
        return out


@dataclass
class Tensor:
    index: int
    name: str
    shape: tuple
    shape_signature: tuple
    dtype: type
    buffer: int
    data: Optional[np.ndarray]  # constant payload or None for activations


@dataclass
class Operator:
    index: int
    op: str
    code: int
    inputs: List[int]
    outputs: List[int]
    options: Dict[str, object] = field(default_factory=dict)


@dataclass
class SubGraph:
    name: str
    tensors: List[Tensor]
    inputs: List[int]
    outputs: List[int]
    operators: List[Operator]

    def producer(self, tensor_index: int) -> Optional[Operator]:
        for op in self.operators:
            if tensor_index in op.outputs:
                return op
        return None

    def consumers(self, tensor_index: int) -> List[Operator]:
        return [op for op in self.operators if tensor_index in op.inputs]


@dataclass
class Model:
    version: int
    description: str
    subgraphs: List[SubGraph]

    @property
    def main(self) -> SubGraph:
        return self.subgraphs[0]


def _options(op_name: str, t: Optional[_Table]) -> Dict[str, object]:
    """Decode the builtin_options table of the ops whose options matter."""
    if t is None:
        return {}
    if op_name == "CONV_2D":
        return {
            "padding": "SAME" if t.scalar(0, "b") == 0 else "VALID",
            "stride_w": t.scalar(1, "i"),
            "stride_h": t.scalar(2, "i"),
            "activation": t.scalar(3, "b"),
            "dilation_w": t.scalar(4, "i", 1),
            "dilation_h": t.scalar(5, "i", 1),
        }
    if op_name in ("FULLY_CONNECTED", "ADD", "MUL", "SUB", "DIV"):
        return {"activation": t.scalar(0, "b")}
    if op_name == "CONCATENATION":
        return {"axis": t.scalar(0, "i"), "activation": t.scalar(1, "b")}
    if op_name == "SOFTMAX":
        return {"beta": t.scalar(0, "f", 1.0)}
    if op_name in ("REDUCE_MAX", "MEAN", "SUM"):
        return {"keep_dims": bool(t.scalar(0, "b"))}
    if op_name == "SPLIT":
        return {"num_splits": t.scalar(0, "i")}
    if op_name == "PACK":
        return {"values_count": t.scalar(0, "i"), "axis": t.scalar(1, "i")}
    if op_name == "GATHER":
        return {"axis": t.scalar(0, "i")}
    if op_name == "STRIDED_SLICE":
        return {
            "begin_mask": t.scalar(0, "i"),
            "end_mask": t.scalar(1, "i"),
            "ellipsis_mask": t.scalar(2, "i"),
            "new_axis_mask": t.scalar(3, "i"),
            "shrink_axis_mask": t.scalar(4, "i"),
        }
    if op_name == "WHILE":
        return {"cond_subgraph": t.scalar(0, "i"), "body_subgraph": t.scalar(1, "i")}
    if op_name == "SQUEEZE":
        return {"squeeze_dims": tuple(int(v) for v in t.vector(0, np.int32))}
    if op_name == "CAST":
        return {"in_type": t.scalar(0, "b"), "out_type": t.scalar(1, "b")}
    return {}


def parse(data: bytes) -> Model:
    """Decode a ``.tflite`` byte string into a :class:`Model`."""
    if len(data) < 8 or data[4:8] != b"TFL3":
        raise FlatBufferError("not a TFL3 flatbuffer")
    root = _Table(data, struct.unpack_from("<I", data, 0)[0])
    version = root.scalar(0, "I")
    opcodes = []
    for oc in root.tables(1):
        code = max(oc.scalar(0, "b"), oc.scalar(3, "i"))
        opcodes.append(code)
    buffers = [b.vector(0, np.uint8) for b in root.tables(4)]

    subgraphs = []
    for sg in root.tables(2):
        tensors = []
        for ti, t in enumerate(sg.tables(0)):
            shape = tuple(int(v) for v in t.vector(0, np.int32))
            sig = tuple(int(v) for v in t.vector(7, np.int32)) or shape
            ttype = t.scalar(1, "b")
            if ttype not in TENSOR_DTYPE:
                raise FlatBufferError(f"unsupported tensor type {ttype}")
            dt = TENSOR_DTYPE[ttype]
            bidx = t.scalar(2, "I")
            raw = buffers[bidx] if bidx < len(buffers) else np.zeros(0, np.uint8)
            payload = None
            if raw.size:
                payload = np.frombuffer(raw.tobytes(), dtype=dt).reshape(shape)
            tensors.append(Tensor(ti, t.string(3), shape, sig, dt, bidx, payload))
        ops = []
        for oi, o in enumerate(sg.tables(3)):
            code = opcodes[o.scalar(0, "I")]
            name = BUILTIN.get(code, f"OP_{code}")
            ops.append(
                Operator(
                    oi, name, code,
                    [int(v) for v in o.vector(1, np.int32)],
                    [int(v) for v in o.vector(2, np.int32)],
                    _options(name, o.table(4)),
                )
            )
        subgraphs.append(
            SubGraph(
                sg.string(4), tensors,
                [int(v) for v in sg.vector(1, np.int32)],
                [int(v) for v in sg.vector(2, np.int32)],
                ops,
            )
        )
    return Model(version, root.string(3), subgraphs)


def load(path: str) -> Model:
    with open(path, "rb") as f:
        return parse(f.read())


def io_details(model: Model) -> tuple:
    """``input_details`` / ``output_details`` in the shape the reference's
    callers consume (``spokestack/wakeword/tflite.py:67-90`` reads
    ``input_details[0]["shape"]``; ``models/tensorflow.py:43-51`` uses
    ``["index"]``)."""

    def one(sg: SubGraph, idx: int) -> dict:
        t = sg.tensors[idx]
        return {
            "name": t.name,
            "index": idx,
            "shape": np.array(t.shape, dtype=np.int32),
            "shape_signature": np.array(t.shape_signature, dtype=np.int32),
            "dtype": t.dtype,
        }

    sg = model.main
    return [one(sg, i) for i in sg.inputs], [one(sg, i) for i in sg.outputs]
