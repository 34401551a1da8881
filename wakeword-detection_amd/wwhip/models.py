"""``TFLiteModel``-shaped executor backed by the HIP library.

The reference wraps one ``.tflite`` file per object and runs it through the TensorFlow-Lite
interpreter (``spokestack/models/tensorflow.py:15-69``).  This class keeps that surface -
``TFLiteModel(model_path, **kwargs)``, ``__call__(*args) -> List[np.ndarray]``,
``.input_details`` / ``.output_details`` - but executes on the MI355X: the file's model
directory (``filter.tflite``, ``encode.tflite``, ``detect.tflite`` side by side, as every
caller of the reference lays them out) is loaded once into an :class:`Engine` and the call is
routed by which of the three graphs the path names.
"""
from __future__ import annotations

import os
import threading
from typing import Any, List

import numpy as np

from . import tflite_reader as R
from .engine import Engine

_TLS = threading.local()  # per host thread: {"engines": {(model dir, device): Engine}}


def engine_for(model_dir: str, device: int = 0) -> Engine:
    """One resident :class:`Engine` per (model directory, device, host thread): an engine enqueues on its thread's own
    context (``_lib.default_context``), which is never shared between threads.  Thread-local, like the context: a thread
    that ends takes its engines (the uploaded weights) with it, and no lock is needed around the table."""
    engines = _TLS.__dict__.setdefault("engines", {})
    key = (os.path.abspath(model_dir), device)
    eng = engines.get(key)
    if eng is None or eng.handle is None:
        eng = Engine(model_dir, device=device)
        engines[key] = eng
    return eng


class TFLiteModel:
    """Drop-in for ``spokestack.models.tensorflow.TFLiteModel`` (same call protocol)."""

    def __init__(self, model_path: str, **kwargs: Any) -> None:
        if not os.path.isfile(model_path):
            raise ValueError(f"Could not open '{model_path}'.")
        self._path = model_path
        self._role = os.path.splitext(os.path.basename(model_path))[0]
        if self._role not in ("filter", "encode", "detect"):
            raise ValueError(f"unsupported model file '{model_path}': expected filter/encode/detect.tflite")
        self._graph = R.load(model_path)
        self._input_details, self._output_details = R.io_details(self._graph)
        self._engine = engine_for(os.path.dirname(model_path) or ".", int(kwargs.get("device", 0)))
        eng = self._engine
        if self._role == "encode":
            # report the true output shape (the CRNN export leaves it unresolved)
            shp = (1, eng.enc_shape[1]) if eng.is_crnn else (1,) + eng.enc_shape
            self._output_details[0]["shape"] = np.array(shp, dtype=np.int32)

    # ---- reference surface ------------------------------------------------------------------
    @property
    def input_details(self) -> List[Any]:
        return self._input_details

    @property
    def output_details(self) -> List[Any]:
        return self._output_details

    def __call__(self, *args) -> List[np.ndarray]:
        if len(args) != len(self._input_details):
            raise ValueError(f"expected {len(self._input_details)} input tensor(s), got {len(args)}")
        x = np.asarray(args[0])
        if x.dtype != np.float32:
            raise ValueError(f"Cannot set tensor: Got value of type {x.dtype} but expected type FLOAT32 for input 0")
        eng = self._engine
        if self._role == "filter":
            if x.ndim != 2 or x.shape[1] != eng.n_bins:
                raise ValueError(f"Cannot set tensor: Dimension mismatch. Got {x.shape} but expected (1, {eng.n_bins})")
            return [eng.filter_apply(x)]
        if self._role == "encode":
            if eng.is_crnn:  # [B, 40, 151, 1] -> time-major windows [B, 151, 40]
                if x.ndim != 4 or x.shape[1:] != (eng.n_mel, eng.window, 1):
                    raise ValueError(f"Cannot set tensor: Dimension mismatch. Got {x.shape} but expected "
                                     f"(1, {eng.n_mel}, {eng.window}, 1)")
                wins = np.ascontiguousarray(np.transpose(x[..., 0], (0, 2, 1)))
            else:
                if x.ndim != 3 or x.shape[1:] != (eng.window, eng.n_mel):
                    raise ValueError(f"Cannot set tensor: Dimension mismatch. Got {x.shape} but expected "
                                     f"(1, {eng.window}, {eng.n_mel})")
                wins = x
            _, enc = eng.forward(wins, want_enc=True)
            return [enc.reshape(enc.shape[0], -1) if eng.is_crnn else enc]
        # detect
        per = eng.enc_shape[0] * eng.enc_shape[1]
        if x.size == 0 or x.size % per != 0 or x.shape[-1] != eng.enc_shape[1]:
            raise ValueError(f"Cannot set tensor: Dimension mismatch. Got {x.shape} for the detect input")
        return [eng.detect(x)]
