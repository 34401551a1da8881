"""Weight extraction from the reference's ``.tflite`` graphs *by wiring*.

The reference keeps its arithmetic inside three TFLite graphs per model
directory (``filter.tflite``, ``encode.tflite``, ``detect.tflite``; loaded at
reference ``spokestack/wakeword/tflite.py:51-59`` and
``utils/evaluate_models.py:30-36``).  The HIP path needs the fp32 parameters
of those graphs in a kernel-friendly packed form.  Tensor indices differ
between exports, so every parameter is located by following producer/consumer
edges (e.g. "the FULLY_CONNECTED whose activation input is the WHILE body's
state argument"), never by index.

The packed form ("blob") is what ``ww_model_load`` in ``include/wwhip.h``
consumes::

    u32 magic 'WWHB' | u32 version | u32 kind | u32 n_sections
    n_sections x { char name[24]; u32 offset_bytes; u32 count_f32 }
    payload (fp32 / i32, 16-byte aligned sections)
"""
from __future__ import annotations

import os
import struct
from dataclasses import dataclass
from typing import Dict, List, Optional, Tuple

import numpy as np

from . import tflite_reader as R

KIND_CRNN = 1
KIND_WAVENET = 2
BLOB_MAGIC = 0x42485757  # 'WWHB' little endian
BLOB_VERSION = 1

HEAD_SIGMOID = 0
HEAD_SOFTMAX = 1


class GraphPatternError(ValueError):
    """The graph does not have the structure of a reference CRNN/Wavenet export."""


# --------------------------------------------------------------------------
# small wiring helpers
# --------------------------------------------------------------------------
_PASS_THROUGH = ("RESHAPE", "BATCH_TO_SPACE_ND", "SPACE_TO_BATCH_ND", "EXPAND_DIMS",
                 "SQUEEZE")


def _const(sg: R.SubGraph, idx: int) -> Optional[np.ndarray]:
    if idx < 0:
        return None
    return sg.tensors[idx].data


def _data_inputs(sg: R.SubGraph, op: R.Operator) -> List[int]:
    """Inputs of ``op`` that are activations (no constant payload)."""
    return [i for i in op.inputs if i >= 0 and sg.tensors[i].data is None]


def _const_inputs(sg: R.SubGraph, op: R.Operator) -> List[int]:
    return [i for i in op.inputs if i >= 0 and sg.tensors[i].data is not None]


def _down(sg: R.SubGraph, t: int, stop_ops: Tuple[str, ...]) -> List[Tuple[R.Operator, int]]:
    """Follow ``t`` forward through shape-only ops; return the first consumers whose
    op is in ``stop_ops`` together with the tensor index they consume."""
    found = []
    frontier = [t]
    seen = set()
    while frontier:
        cur = frontier.pop()
        if cur in seen:
            continue
        seen.add(cur)
        for op in sg.consumers(cur):
            if op.op in stop_ops:
                found.append((op, cur))
            elif op.op in _PASS_THROUGH and op.inputs[0] == cur:
                frontier.append(op.outputs[0])
    found.sort(key=lambda p: p[0].index)
    return found


def _up(sg: R.SubGraph, t: int) -> Tuple[Optional[R.Operator], int]:
    """Walk backwards from ``t`` over shape-only ops to the op that computed it."""
    cur = t
    while True:
        op = sg.producer(cur)
        if op is None:
            return None, cur
        if op.op in _PASS_THROUGH:
            cur = op.inputs[0]
            continue
        return op, cur


# --------------------------------------------------------------------------
# filter.tflite  (reference wakeword/tflite.py:183-184, graph: SURVEY A1)
# --------------------------------------------------------------------------
@dataclass
class FilterParams:
    weight: np.ndarray  # [n_mel, n_bins]  (TFLite FULLY_CONNECTED layout [out,in])
    bias: np.ndarray  # [n_mel]
    floor: float  # MAXIMUM constant
    log_offset: float  # value added after LOG  (SUB of a negative constant)
    scale: float  # final MUL

    @property
    def n_bins(self) -> int:
        return int(self.weight.shape[1])

    @property
    def n_mel(self) -> int:
        return int(self.weight.shape[0])


def extract_filter(model: R.Model) -> FilterParams:
    sg = model.main
    ops = {o.op: o for o in sg.operators}
    for need in ("FULLY_CONNECTED", "MAXIMUM", "LOG", "SUB", "MUL"):
        if need not in ops:
            raise GraphPatternError(f"filter graph lacks {need}")
    fc = ops["FULLY_CONNECTED"]
    if fc.inputs[0] != sg.inputs[0]:
        raise GraphPatternError("filter FULLY_CONNECTED is not fed by the graph input")
    w = _const(sg, fc.inputs[1]).astype(np.float32)
    b = _const(sg, fc.inputs[2]) if len(fc.inputs) > 2 else None
    b = np.zeros(w.shape[0], np.float32) if b is None else b.astype(np.float32)
    floor = float(_const(sg, _const_inputs(sg, ops["MAXIMUM"])[0]).ravel()[0])
    sub = ops["SUB"]
    # y = log(..) - c  with the constant as second operand
    if _const(sg, sub.inputs[1]) is None:
        raise GraphPatternError("filter SUB constant is not the subtrahend")
    log_offset = -float(_const(sg, sub.inputs[1]).ravel()[0])
    scale = float(_const(sg, _const_inputs(sg, ops["MUL"])[0]).ravel()[0])
    return FilterParams(w, b, floor, log_offset, scale)


# --------------------------------------------------------------------------
# CRNN encode/detect  (reference wwdetect/CRNN/model.py:21-56, graph: SURVEY A2)
# --------------------------------------------------------------------------
@dataclass
class GruDir:
    w_x: np.ndarray  # [3H, in]  gate order z, r, h
    b_x: np.ndarray  # [3H]
    w_h: np.ndarray  # [3H, H]
    b_h: np.ndarray  # [3H]


@dataclass
class CrnnParams:
    n_mel: int
    n_frames: int
    conv_w: np.ndarray  # [C, kf, kt]
    conv_b: np.ndarray  # [C]
    stride_f: int
    stride_t: int
    pad_f: Tuple[int, int]
    pad_t: Tuple[int, int]
    out_f: int
    out_t: int
    gru1: Tuple[GruDir, GruDir]  # (forward, backward)
    gru2: Tuple[GruDir, GruDir]
    head_w1: np.ndarray  # [64, 64]
    head_b1: np.ndarray
    head_w2: np.ndarray  # [n_out, 64]
    head_b2: np.ndarray
    head_kind: int  # HEAD_SIGMOID | HEAD_SOFTMAX

    @property
    def n_out(self) -> int:
        return int(self.head_w2.shape[0])

    @property
    def units(self) -> int:
        return int(self.gru1[0].w_h.shape[1])


def _same_pad(n_in: int, k: int, s: int) -> Tuple[int, Tuple[int, int]]:
    n_out = -(-n_in // s)
    total = max((n_out - 1) * s + k - n_in, 0)
    return n_out, (total // 2, total - total // 2)


def _gru_from_body(body: R.SubGraph) -> GruDir:
    state_arg = body.inputs[3]
    fcs = [o for o in body.operators if o.op == "FULLY_CONNECTED"]
    if len(fcs) != 2:
        raise GraphPatternError("GRU body must hold exactly two FULLY_CONNECTED ops")
    rec = [o for o in fcs if o.inputs[0] == state_arg]
    inp = [o for o in fcs if o.inputs[0] != state_arg]
    if len(rec) != 1 or len(inp) != 1:
        raise GraphPatternError("cannot tell recurrent from input projection")
    prod = body.producer(inp[0].inputs[0])
    if prod is None or prod.op != "GATHER":
        raise GraphPatternError("input projection is not fed by GATHER(x, t)")
    # gate algebra check (reset_after GRU, order z|r|h):  h' = z*h + (1-z)*tanh(xh + r*hh)
    names = [o.op for o in body.operators]
    if names.count("LOGISTIC") != 2 or names.count("TANH") != 1:
        raise GraphPatternError("unexpected gate structure in GRU body")
    # the LOGISTIC whose result multiplies the state is z; the one that multiplies the
    # recurrent split is r.  Verify z is split 0 and r is split 1.
    split_h = [o for o in body.operators if o.op == "SPLIT" and o.inputs[1] == rec[0].outputs[0]][0]
    split_x = [o for o in body.operators if o.op == "SPLIT" and o.inputs[1] == inp[0].outputs[0]][0]
    for lg in (o for o in body.operators if o.op == "LOGISTIC"):
        add = body.producer(lg.inputs[0])
        gate_idx = split_x.outputs.index([i for i in add.inputs if i in split_x.outputs][0])
        muls = [o for o in body.consumers(lg.outputs[0]) if o.op == "MUL"]
        if any(state_arg in m.inputs for m in muls):
            if gate_idx != 0:
                raise GraphPatternError("update gate is not the first split")
        elif any(split_h.outputs[2] in m.inputs for m in muls):
            if gate_idx != 1:
                raise GraphPatternError("reset gate is not the second split")
        else:
            raise GraphPatternError("LOGISTIC output feeds neither h nor the candidate")
    return GruDir(
        _const(body, inp[0].inputs[1]).astype(np.float32),
        _const(body, inp[0].inputs[2]).astype(np.float32),
        _const(body, rec[0].inputs[1]).astype(np.float32),
        _const(body, rec[0].inputs[2]).astype(np.float32),
    )


def extract_crnn(encode: R.Model, detect: R.Model) -> CrnnParams:
    sg = encode.main
    convs = [o for o in sg.operators if o.op == "CONV_2D"]
    if len(convs) != 1 or convs[0].inputs[0] != sg.inputs[0]:
        raise GraphPatternError("CRNN encoder must start with one CONV_2D on the input")
    conv = convs[0]
    if conv.options["padding"] not in ("SAME", "VALID") or conv.options["activation"] != 1:
        raise GraphPatternError("CRNN conv must be SAME or VALID with fused ReLU")
    in_shape = sg.tensors[sg.inputs[0]].shape  # [1, mel, frames, 1]
    n_mel, n_frames = int(in_shape[1]), int(in_shape[2])
    w = _const(sg, conv.inputs[1]).astype(np.float32)  # [C, kf, kt, 1]
    if w.shape[3] != 1:
        raise GraphPatternError("CRNN conv expects one input channel")
    sf, st = int(conv.options["stride_h"]), int(conv.options["stride_w"])
    if conv.options["padding"] == "SAME":
        out_f, pad_f = _same_pad(n_mel, w.shape[1], sf)
        out_t, pad_t = _same_pad(n_frames, w.shape[2], st)
    else:  # the older export under utils/CRNN_files/*_old.tflite: 20x5 kernel, stride 8x2, no padding
        out_f, pad_f = (n_mel - w.shape[1]) // sf + 1, (0, 0)
        out_t, pad_t = (n_frames - w.shape[2]) // st + 1, (0, 0)
    if tuple(sg.tensors[conv.outputs[0]].shape[1:3]) != (out_f, out_t):
        raise GraphPatternError("conv output shape differs from the padding arithmetic")

    # feature order check: TRANSPOSE [0,2,1,3] then RESHAPE -> feature = f*C + c
    tr = sg.consumers(conv.outputs[0])
    if len(tr) != 1 or tr[0].op != "TRANSPOSE" or list(_const(sg, tr[0].inputs[1])) != [0, 2, 1, 3]:
        raise GraphPatternError("expected TRANSPOSE perm [0,2,1,3] after the conv")

    whiles = [o for o in sg.operators if o.op == "WHILE"]
    if len(whiles) != 4:
        raise GraphPatternError("expected 4 WHILE loops (2 layers x 2 directions)")
    layers: Dict[int, Dict[bool, GruDir]] = {1: {}, 2: {}}
    for wop in whiles:
        body = encode.subgraphs[wop.options["body_subgraph"]]
        g = _gru_from_body(body)
        seq_in = wop.inputs[5]
        prod = sg.producer(seq_in)
        backward = prod is not None and prod.op == "REVERSE_V2"
        layer = 1 if g.w_x.shape[1] == out_f * w.shape[0] else 2
        if backward in layers[layer]:
            raise GraphPatternError("duplicate direction in a GRU layer")
        layers[layer][backward] = g
    for l in (1, 2):
        if set(layers[l]) != {False, True}:
            raise GraphPatternError(f"GRU layer {l} lacks a direction")
    units = layers[1][False].w_h.shape[1]
    if layers[2][False].w_x.shape[1] != 2 * units:
        raise GraphPatternError("layer-2 input width is not 2*units")

    # output order: CONCATENATION(fwd_last, bwd_last)
    out_prod = sg.producer(sg.outputs[0])
    if out_prod.op != "CONCATENATION":
        raise GraphPatternError("encoder output is not a CONCATENATION")

    dsg = detect.main
    fcs = [o for o in dsg.operators if o.op == "FULLY_CONNECTED"]
    if len(fcs) != 2 or fcs[0].inputs[0] != dsg.inputs[0] or fcs[1].inputs[0] != fcs[0].outputs[0]:
        raise GraphPatternError("detect graph must be FC -> FC -> activation")
    if fcs[0].options["activation"] != 1 or fcs[1].options["activation"] != 0:
        raise GraphPatternError("detect head activations differ from ReLU / linear")
    last = dsg.producer(dsg.outputs[0])
    if last.op == "LOGISTIC":
        head_kind = HEAD_SIGMOID
    elif last.op == "SOFTMAX":
        head_kind = HEAD_SOFTMAX
        if abs(float(last.options.get("beta", 1.0)) - 1.0) > 0:
            raise GraphPatternError("softmax beta != 1")
    else:
        raise GraphPatternError(f"unknown head activation {last.op}")
    return CrnnParams(
        n_mel, n_frames,
        np.ascontiguousarray(w[:, :, :, 0]), _const(sg, conv.inputs[2]).astype(np.float32),
        sf, st, pad_f, pad_t, out_f, out_t,
        (layers[1][False], layers[1][True]), (layers[2][False], layers[2][True]),
        _const(dsg, fcs[0].inputs[1]).astype(np.float32), _const(dsg, fcs[0].inputs[2]).astype(np.float32),
        _const(dsg, fcs[1].inputs[1]).astype(np.float32), _const(dsg, fcs[1].inputs[2]).astype(np.float32),
        head_kind,
    )


# --------------------------------------------------------------------------
# Wavenet encode/detect (reference wwdetect/wavenet/wavenet_model.py:11-128)
# --------------------------------------------------------------------------
@dataclass
class WaveBlock:
    dilation: int
    bn_scale: np.ndarray  # [C]
    bn_shift: np.ndarray  # [C]
    w_sig: np.ndarray  # [3, C_in, C_out] tap k reads u[t-(2-k)d]
    b_sig: np.ndarray
    w_tanh: np.ndarray
    b_tanh: np.ndarray
    w_res: Optional[np.ndarray]  # [C_in, C_out] or None (last block)
    b_res: Optional[np.ndarray]
    w_skip: np.ndarray  # [C_in, S]
    b_skip: np.ndarray


@dataclass
class WavenetParams:
    n_frames: int
    n_mel: int
    w_in: np.ndarray  # [n_mel, C]
    b_in: np.ndarray
    blocks: List[WaveBlock]
    skip_order: List[int]  # order in which skip tensors are summed
    det_w1: np.ndarray  # [S, S]  (in, out)
    det_b1: np.ndarray
    det_w2: np.ndarray  # [S, n_out]
    det_b2: np.ndarray

    @property
    def channels(self) -> int:
        return int(self.w_in.shape[1])

    @property
    def skip_channels(self) -> int:
        return int(self.det_w1.shape[0])


def _conv1x_weights(sg: R.SubGraph, conv: R.Operator) -> Tuple[np.ndarray, np.ndarray]:
    if conv.options["padding"] != "VALID" or conv.options["stride_w"] != 1:
        raise GraphPatternError("Wavenet convs must be VALID stride-1")
    if conv.options["dilation_w"] != 1 or conv.options["activation"] != 0:
        raise GraphPatternError("Wavenet convs must be undilated (S2B form) without fused act")
    w = _const(sg, conv.inputs[1]).astype(np.float32)  # [out, 1, kw, in]
    if w.shape[1] != 1:
        raise GraphPatternError("Wavenet conv kernel height != 1")
    b = _const(sg, conv.inputs[2]).astype(np.float32)
    return np.ascontiguousarray(w[:, 0].transpose(1, 2, 0)), b  # [kw, in, out]


def _bias_chain(sg: R.SubGraph, t: int, stop: Tuple[str, ...]):
    """From tensor ``t`` walk forward over shape ops and const-ADDs until an op in
    ``stop`` (or a fused-activation ADD) is met.  Returns (extra_bias, relu, tensor, op)."""
    extra = None
    cur = t
    while True:
        cons = sg.consumers(cur)
        if len(cons) != 1:
            return extra, False, cur, None
        op = cons[0]
        if op.op in _PASS_THROUGH:
            cur = op.outputs[0]
            continue
        if op.op == "ADD" and len(_const_inputs(sg, op)) == 1:
            c = _const(sg, _const_inputs(sg, op)[0]).astype(np.float32).ravel()
            extra = c if extra is None else extra + c
            cur = op.outputs[0]
            if op.options.get("activation", 0) == 1:
                return extra, True, cur, op
            continue
        if op.op in stop:
            return extra, False, cur, op
        return extra, False, cur, op


def extract_wavenet(encode: R.Model, detect: R.Model) -> WavenetParams:
    sg = encode.main
    in_t = sg.inputs[0]
    n_frames, n_mel = (int(v) for v in sg.tensors[in_t].shape[1:3])

    first = _down(sg, in_t, ("CONV_2D",))
    if len(first) != 1:
        raise GraphPatternError("Wavenet input must feed exactly one conv")
    w_in, b_in = _conv1x_weights(sg, first[0][0])
    if w_in.shape[0] != 1:
        raise GraphPatternError("Wavenet input conv must be 1x1")
    extra, relu, x, _ = _bias_chain(sg, first[0][0].outputs[0], ())
    if not relu:
        raise GraphPatternError("Wavenet input conv lacks ReLU")
    b_in = b_in + (extra if extra is not None else 0)
    channels = w_in.shape[2]

    blocks: List[WaveBlock] = []
    skip_tensor_of_block: Dict[int, int] = {}
    while True:
        # x feeds: MUL(const) [BN scale] and ADD(res, x) [residual]; the last block has no residual
        muls = [o for o in sg.consumers(x) if o.op == "MUL" and len(_const_inputs(sg, o)) == 1]
        if len(muls) != 1:
            break
        scale = _const(sg, _const_inputs(sg, muls[0])[0]).astype(np.float32).ravel()
        adds = sg.consumers(muls[0].outputs[0])
        if len(adds) != 1 or adds[0].op != "ADD":
            raise GraphPatternError("BN scale not followed by shift")
        shift = _const(sg, _const_inputs(sg, adds[0])[0]).astype(np.float32).ravel()
        pads = sg.consumers(adds[0].outputs[0])
        if len(pads) != 1 or pads[0].op != "PAD":
            raise GraphPatternError("BN not followed by causal PAD")
        padv = _const(sg, pads[0].inputs[1])
        if padv.shape != (3, 2) or padv[0].any() or padv[2].any() or padv[1][1] != 0:
            raise GraphPatternError("PAD is not a pure left time pad")
        left = int(padv[1][0])
        dilation = left // 2
        if dilation * 2 != left:
            raise GraphPatternError("odd causal pad")
        s2b = [o for o in sg.consumers(pads[0].outputs[0]) if o.op == "SPACE_TO_BATCH_ND"]
        if s2b:
            if int(_const(sg, s2b[0].inputs[1]).ravel()[0]) != dilation:
                raise GraphPatternError("S2B block size != dilation")
        elif dilation != 1:
            raise GraphPatternError("dilated block without SPACE_TO_BATCH_ND")
        gates = _down(sg, pads[0].outputs[0], ("CONV_2D",))
        if len(gates) != 2:
            raise GraphPatternError("expected two gate convs per block")
        gw = {}
        act_out = {}
        for conv, _ in gates:
            w, b = _conv1x_weights(sg, conv)
            if w.shape != (3, channels, channels):
                raise GraphPatternError("gate conv has unexpected shape")
            extra, _, t, op = _bias_chain(sg, conv.outputs[0], ("LOGISTIC", "TANH"))
            if op is None or op.op not in ("LOGISTIC", "TANH"):
                raise GraphPatternError("gate conv not followed by LOGISTIC/TANH")
            if op.op in gw:
                raise GraphPatternError("both gate convs share an activation")
            gw[op.op] = (w, b + (extra if extra is not None else 0))
            act_out[op.op] = op.outputs[0]
        mm = [o for o in sg.consumers(act_out["TANH"]) if o.op == "MUL"]
        if len(mm) != 1 or act_out["LOGISTIC"] not in mm[0].inputs:
            raise GraphPatternError("gates are not multiplied together")
        m = mm[0].outputs[0]
        w_res = b_res = None
        w_skip = b_skip = None
        x_next = None
        for conv, _ in _down(sg, m, ("CONV_2D",)):
            w, b = _conv1x_weights(sg, conv)
            if w.shape[0] != 1:
                raise GraphPatternError("res/skip conv must be 1x1")
            extra, relu, t, _ = _bias_chain(sg, conv.outputs[0], ())
            if not relu:
                raise GraphPatternError("res/skip conv lacks ReLU")
            b = b + (extra if extra is not None else 0)
            nxt = [o for o in sg.consumers(t) if o.op == "ADD" and x in o.inputs]
            if nxt:
                if w_res is not None:
                    raise GraphPatternError("two residual convs in one block")
                w_res, b_res = w[0], b
                x_next = nxt[0].outputs[0]
            else:
                if w_skip is not None:
                    raise GraphPatternError("two skip convs in one block")
                w_skip, b_skip = w[0], b
                skip_tensor_of_block[len(blocks)] = t
        if w_skip is None:
            raise GraphPatternError("block without skip conv")
        blocks.append(WaveBlock(dilation, scale, shift, gw["LOGISTIC"][0], gw["LOGISTIC"][1],
                                gw["TANH"][0], gw["TANH"][1], w_res, b_res, w_skip, b_skip))
        if x_next is None:
            break
        x = x_next
    if not blocks:
        raise GraphPatternError("no Wavenet blocks found")

    # order of the skip summation: unroll the ADD chain that produces the output
    block_of_tensor = {t: b for b, t in skip_tensor_of_block.items()}
    order: List[int] = []

    def unroll(t: int) -> None:
        if t in block_of_tensor:
            order.append(block_of_tensor[t])
            return
        op = sg.producer(t)
        if op is None or op.op != "ADD" or len(op.inputs) != 2:
            raise GraphPatternError("skip sum is not a chain of ADDs")
        unroll(op.inputs[0])
        unroll(op.inputs[1])

    unroll(sg.outputs[0])
    if sorted(order) != list(range(len(blocks))):
        raise GraphPatternError("skip sum does not cover every block exactly once")

    dsg = detect.main
    relu = sg_first = None
    cons = dsg.consumers(dsg.inputs[0])
    if len(cons) != 1 or cons[0].op != "RELU":
        raise GraphPatternError("Wavenet detect must start with RELU")
    convs = [o for o in dsg.operators if o.op == "CONV_2D"]
    if len(convs) != 2:
        raise GraphPatternError("Wavenet detect must hold two 1x1 convs")
    w1, b1 = _conv1x_weights(dsg, convs[0])
    extra, relu, t1, _ = _bias_chain(dsg, convs[0].outputs[0], ("CONV_2D",))
    b1 = b1 + (extra if extra is not None else 0)
    if not relu:
        raise GraphPatternError("first detect conv lacks ReLU")
    w2, b2 = _conv1x_weights(dsg, convs[1])
    extra, _, t2, op = _bias_chain(dsg, convs[1].outputs[0], ("REDUCE_MAX",))
    b2 = b2 + (extra if extra is not None else 0)
    if op is None or op.op != "REDUCE_MAX":
        raise GraphPatternError("detect lacks REDUCE_MAX over time")
    sm = dsg.producer(dsg.outputs[0])
    if sm.op != "SOFTMAX" or abs(float(sm.options.get("beta", 1.0)) - 1.0) > 0:
        raise GraphPatternError("detect must end in SOFTMAX(beta=1)")
    return WavenetParams(n_frames, n_mel, w_in[0], b_in, blocks, order, w1[0], b1, w2[0], b2)


# --------------------------------------------------------------------------
# model directory -> params -> blob
# --------------------------------------------------------------------------
@dataclass
class ModelBundle:
    kind: int
    filt: FilterParams
    crnn: Optional[CrnnParams] = None
    wavenet: Optional[WavenetParams] = None
    encode_io: tuple = ()
    detect_io: tuple = ()
    filter_io: tuple = ()

    @property
    def window(self) -> int:
        return self.crnn.n_frames if self.kind == KIND_CRNN else self.wavenet.n_frames

    @property
    def n_out(self) -> int:
        return self.crnn.n_out if self.kind == KIND_CRNN else int(self.wavenet.det_w2.shape[1])

    @property
    def posterior_index(self) -> int:
        """Quirk C1 of SURVEY: width-1 head -> element 0, width-2 head -> element 1
        (reference wakeword/tflite.py:228-231 vs evaluate_models.py:80,86)."""
        return 0 if self.n_out == 1 else 1


def detect_kind(encode: R.Model) -> int:
    ops = {o.op for o in encode.main.operators}
    return KIND_CRNN if "WHILE" in ops else KIND_WAVENET


def load_model_dir(model_dir: str) -> ModelBundle:
    paths = {n: os.path.join(model_dir, f"{n}.tflite") for n in ("filter", "encode", "detect")}
    for p in paths.values():
        if not os.path.isfile(p):
            raise FileNotFoundError(p)
    f = R.load(paths["filter"])
    e = R.load(paths["encode"])
    d = R.load(paths["detect"])
    kind = detect_kind(e)
    b = ModelBundle(kind, extract_filter(f))
    if kind == KIND_CRNN:
        b.crnn = extract_crnn(e, d)
    else:
        b.wavenet = extract_wavenet(e, d)
    b.filter_io, b.encode_io, b.detect_io = R.io_details(f), R.io_details(e), R.io_details(d)
    return b


def quantize_fp16(bundle: ModelBundle) -> ModelBundle:
    """The float16 post-training-quantised variant the reference also builds and evaluates
    (``wwdetect/CRNN/model.py:165-179``, ``wavenet_model.py:149-163``: ``supported_types = [tf.float16]``;
    compared in ``utils/evaluate_tf_lite_opts.py:107-125``): every constant of encode / detect is stored as
    float16 and dequantised to float32 when the model is loaded - the arithmetic stays float32.  The
    ``-quant.tflite`` files are not shipped, so the same rounding is applied to the extracted constants here
    (filter.tflite is a separate, unquantised graph and is left alone)."""
    import copy
    import dataclasses

    def rnd(obj):
        if isinstance(obj, np.ndarray) and obj.dtype == np.float32:
            return obj.astype(np.float16).astype(np.float32)
        if dataclasses.is_dataclass(obj) and not isinstance(obj, type):
            return dataclasses.replace(obj, **{f.name: rnd(getattr(obj, f.name)) for f in dataclasses.fields(obj)})
        if isinstance(obj, tuple):
            return tuple(rnd(x) for x in obj)
        if isinstance(obj, list):
            return [rnd(x) for x in obj]
        return obj

    out = copy.copy(bundle)
    out.crnn = rnd(bundle.crnn) if bundle.crnn is not None else None
    out.wavenet = rnd(bundle.wavenet) if bundle.wavenet is not None else None
    return out


def _sections(bundle: ModelBundle) -> List[Tuple[str, np.ndarray]]:
    f = bundle.filt
    sec: List[Tuple[str, np.ndarray]] = [
        ("filter.meta", np.array([f.n_mel, f.n_bins], np.int32)),
        ("filter.consts", np.array([f.floor, f.log_offset, f.scale], np.float32)),
        ("filter.w", f.weight),
        ("filter.b", f.bias),
    ]
    if bundle.kind == KIND_CRNN:
        c = bundle.crnn
        sec += [
            ("crnn.meta", np.array([c.n_mel, c.n_frames, c.conv_w.shape[0], c.conv_w.shape[1],
                                    c.conv_w.shape[2], c.stride_f, c.stride_t, c.pad_f[0], c.pad_t[0],
                                    c.out_f, c.out_t, c.units, c.n_out, c.head_kind], np.int32)),
            ("crnn.conv_w", c.conv_w), ("crnn.conv_b", c.conv_b),
        ]
        for li, layer in ((1, c.gru1), (2, c.gru2)):
            for dn, g in zip(("f", "b"), layer):
                sec += [(f"crnn.g{li}{dn}.wx", g.w_x), (f"crnn.g{li}{dn}.bx", g.b_x),
                        (f"crnn.g{li}{dn}.wh", g.w_h), (f"crnn.g{li}{dn}.bh", g.b_h)]
        sec += [("crnn.head_w1", c.head_w1), ("crnn.head_b1", c.head_b1),
                ("crnn.head_w2", c.head_w2), ("crnn.head_b2", c.head_b2)]
    else:
        w = bundle.wavenet
        nb = len(w.blocks)
        sec += [
            ("wave.meta", np.array([w.n_frames, w.n_mel, w.channels, w.skip_channels, nb,
                                    int(w.det_w2.shape[1])], np.int32)),
            ("wave.dilations", np.array([b.dilation for b in w.blocks], np.int32)),
            ("wave.skip_order", np.array(w.skip_order, np.int32)),
            ("wave.has_res", np.array([b.w_res is not None for b in w.blocks], np.int32)),
            ("wave.w_in", w.w_in), ("wave.b_in", w.b_in),
            ("wave.bn_scale", np.stack([b.bn_scale for b in w.blocks])),
            ("wave.bn_shift", np.stack([b.bn_shift for b in w.blocks])),
            ("wave.w_sig", np.stack([b.w_sig for b in w.blocks])),
            ("wave.b_sig", np.stack([b.b_sig for b in w.blocks])),
            ("wave.w_tanh", np.stack([b.w_tanh for b in w.blocks])),
            ("wave.b_tanh", np.stack([b.b_tanh for b in w.blocks])),
            ("wave.w_res", np.stack([b.w_res if b.w_res is not None else
                                     np.zeros((w.channels, w.channels), np.float32) for b in w.blocks])),
            ("wave.b_res", np.stack([b.b_res if b.b_res is not None else
                                     np.zeros(w.channels, np.float32) for b in w.blocks])),
            ("wave.w_skip", np.stack([b.w_skip for b in w.blocks])),
            ("wave.b_skip", np.stack([b.b_skip for b in w.blocks])),
            ("wave.det_w1", w.det_w1), ("wave.det_b1", w.det_b1),
            ("wave.det_w2", w.det_w2), ("wave.det_b2", w.det_b2),
        ]
    return sec


def pack_blob(bundle: ModelBundle) -> bytes:
    """Serialise a :class:`ModelBundle` into the byte layout documented in
    ``include/wwhip.h`` (``ww_model_load``)."""
    sec = _sections(bundle)
    header = 16 + 32 * len(sec)
    header = (header + 15) // 16 * 16
    table = b""
    payload = b""
    for name, arr in sec:
        arr = np.ascontiguousarray(arr)
        if arr.dtype not in (np.float32, np.int32):
            raise TypeError(f"section {name} has dtype {arr.dtype}")
        raw = arr.tobytes()
        off = header + len(payload)
        nm = name.encode("ascii")
        if len(nm) > 23:
            raise ValueError(name)
        table += nm.ljust(24, b"\0") + struct.pack("<II", off, arr.size)
        payload += raw + b"\0" * (-len(raw) % 16)
    head = struct.pack("<IIII", BLOB_MAGIC, BLOB_VERSION, bundle.kind, len(sec)) + table
    head = head.ljust(header, b"\0")
    return head + payload


def unpack_blob(blob: bytes) -> Dict[str, np.ndarray]:
    """Inverse of :func:`pack_blob` for tests and the CPU oracle (sections come back
    flat; int sections are those named ``*.meta`` / dilations / order / has_res)."""
    magic, ver, kind, n = struct.unpack_from("<IIII", blob, 0)
    if magic != BLOB_MAGIC or ver != BLOB_VERSION:
        raise ValueError("bad blob header")
    out: Dict[str, np.ndarray] = {"__kind__": np.array([kind], np.int32)}
    int_names = ("meta", "dilations", "skip_order", "has_res")
    for i in range(n):
        base = 16 + 32 * i
        name = blob[base : base + 24].split(b"\0", 1)[0].decode("ascii")
        off, cnt = struct.unpack_from("<II", blob, base + 24)
        dt = np.int32 if name.rsplit(".", 1)[-1] in int_names else np.float32
        out[name] = np.frombuffer(blob, dtype=dt, count=cnt, offset=off)
    return out
