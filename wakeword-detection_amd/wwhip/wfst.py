"""Superframe shortest-path smoothing (``wwdetect/wfst.py:17-71``) without pynini.

The reference turns a superframe of T (=10) per-step posteriors ``[p_other, p_wakeword]`` into a
2-state lattice and asks OpenFst for the single shortest path (tropical semiring): start arcs cost
``ln 2 - ln p[0][s]``, an arc into state ``s`` at step t costs ``-ln p[t][s]`` minus a stay bonus of
1 when the state does not change; the detector fires when the best path visits ``wakeword``
(``utils/CRNN_files/tflite.py:252-263``).  That is a 2 x T Viterbi recursion - ``ww_superframe_smooth``
runs one GPU thread per superframe, so a whole evaluation set (or every stream of a
``WakewordBank``) is smoothed in one launch.

``smooth(posterior_probs)`` keeps the reference's call shape and returns the label sequence the
reference prints (``smoothed.stringify(token_type=state_table)``), e.g. ``"other other wakeword"``.
"""
from __future__ import annotations

import ctypes as C
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

LABELS = ("other", "wakeword")


def smooth_batch(posterior_probs: np.ndarray, stay_bonus: float = 1.0, ctx: Optional[_lib.Context] = None,
                 device_log: bool = False) -> Tuple[np.ndarray, np.ndarray]:
    """``posterior_probs [n, T, 2]`` -> ``(paths uint8 [n, T], wake bool [n])``.  The costs are
    ``-np.log(p)`` computed here in float32, exactly as the reference computes them, unless
    ``device_log`` asks the kernel to take the logarithm (same result up to 1 ulp of the cost)."""
    p = np.ascontiguousarray(posterior_probs, dtype=np.float32)
    if p.ndim != 3 or p.shape[2] != 2:
        raise ValueError("posterior_probs must have shape [n, T, 2]")
    n, T, _ = p.shape
    ctx = ctx or _lib.default_context()
    with np.errstate(divide="ignore"):
        arr = p if device_log else np.ascontiguousarray(-np.log(p))
    path = np.empty((n, T), np.uint8)
    wake = np.empty(n, np.uint8)
    _lib.raise_for(_lib.load().ww_superframe_smooth(ctx.handle, _lib.ptr(arr), n, T, C.c_float(stay_bonus),
                                                    0 if device_log else 1, _lib.ptr(path), _lib.ptr(wake)), ctx.handle)
    return path, wake.astype(bool)


def smooth(posterior_probs: Sequence[Sequence[float]]) -> str:
    """Reference signature: one superframe ``[[p_other, p_wakeword], ...]`` -> space-separated labels."""
    path, _ = smooth_batch(np.asarray(posterior_probs, dtype=np.float32)[None])
    return " ".join(LABELS[s] for s in path[0])


class SuperframeDetector:
    """The trigger logic of ``utils/CRNN_files/tflite.py:252-263``: collect ``superframe_len`` posteriors,
    smooth, fire if the best path contains 'wakeword', start over."""

    def __init__(self, superframe_len: int = 10) -> None:
        self.superframe_len = superframe_len
        self.superframe: List[List[float]] = []

    def push(self, posterior: float) -> bool:
        post_ww = np.float32(posterior)
        post_other = np.abs(post_ww - np.float32(1))
        self.superframe.append([post_other, post_ww])
        if len(self.superframe) < self.superframe_len:
            return False
        fired = "wakeword" in smooth(self.superframe).split()
        self.superframe = []
        return fired

    def reset(self) -> None:
        self.superframe = []
