"""ORACLE (test infrastructure only) - ctypes front for ``oracle/ww_oracle.c``.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this.  Parity status: see the header of ``ww_oracle.c`` (front end pinned by
reference-generated fixtures; model arithmetic "parity unpinned" - no TFLite here).

The shared object is built by ``oracle/Makefile`` (``__graft_entry__.build()`` calls it).
A host-tuned variant (``-march=native``) is built on first use when gcc is present so
that the CPU baseline is not handicapped by a portable build; the portable
``libwworacle.so`` is the fallback.
"""
from __future__ import annotations

import ctypes as C
import hashlib
import os
import platform
import subprocess
from typing import Optional, Tuple

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB: Optional[C.CDLL] = None
_LIB_PATH = ""


def _cpu_tag() -> str:
    flags = ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    flags = line
                    break
    except OSError:
        pass
    return hashlib.sha1((platform.machine() + flags).encode()).hexdigest()[:10]


def build(native: bool = True) -> str:
    """Compile the oracle; returns the path of the library to load."""
    src = os.path.join(_HERE, "ww_oracle.c")
    portable = os.path.join(_HERE, "libwworacle.so")
    base = ["gcc", "-O3", "-fopenmp", "-fPIC", "-std=gnu11", "-fno-fast-math", "-shared"]
    if not os.path.isfile(portable) or os.path.getmtime(portable) < os.path.getmtime(src):
        subprocess.run(base + ["-march=x86-64-v3", "-o", portable, src, "-lm"], check=True)
    if not native:
        return portable
    tuned = os.path.join(_HERE, f"libwworacle.{_cpu_tag()}.so")
    if not os.path.isfile(tuned) or os.path.getmtime(tuned) < os.path.getmtime(src):
        try:
            subprocess.run(base + ["-march=native", "-o", tuned, src, "-lm"], check=True,
                           capture_output=True, timeout=120)
        except (OSError, subprocess.SubprocessError):
            return portable
    return tuned


def lib() -> C.CDLL:
    global _LIB, _LIB_PATH
    if _LIB is not None:
        return _LIB
    path = os.path.join(_HERE, "libwworacle.so")
    try:
        path = build(native=True)
    except (OSError, subprocess.SubprocessError):
        if not os.path.isfile(path):
            raise
    L = C.CDLL(path)
    _LIB_PATH = path
    i64, i32, f32 = C.c_int64, C.c_int, C.c_float
    vp, sz = C.c_void_p, C.c_size_t
    L.wwo_num_threads.restype = i32
    L.wwo_set_threads.argtypes = [i32]
    L.wwo_num_frames.restype = i64
    L.wwo_num_frames.argtypes = [i64, i32]
    L.wwo_logmel.argtypes = [vp, sz, vp, i64, f32, i32, f32, i32, vp, i32, vp, vp]
    L.wwo_logmel_f32.argtypes = [vp, sz, vp, i64, f32, i32, vp, vp]
    L.wwo_stft_mag.argtypes = [vp, i64, vp]
    L.wwo_crnn_forward.argtypes = [vp, sz, vp, i32, vp, vp]
    L.wwo_wavenet_forward.argtypes = [vp, sz, vp, i32, vp, vp]
    L.wwo_slide_forward.argtypes = [vp, sz, vp, i64, i32, vp, vp]
    L.wwo_smooth.argtypes = [vp, i64, i32, vp]
    L.wwo_far_frr.argtypes = [vp, i64, vp, i64, vp, i32, C.c_double, C.c_double, vp, vp, vp]
    _LIB = L
    return L


def lib_path() -> str:
    lib()
    return _LIB_PATH


def num_threads() -> int:
    return int(lib().wwo_num_threads())


def set_threads(n: int) -> int:
    """Fix the OpenMP team size (the GPU boxes expose far more logical CPUs than the job's
    share; oversubscribing them makes the baseline slower, not faster)."""
    lib().wwo_set_threads(int(n))
    return num_threads()


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _chk(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with {rc}")


class CpuOracle:
    """CPU restatement bound to one packed weight blob (``wwhip.weights.pack_blob``)."""

    def __init__(self, blob: bytes) -> None:
        self._blob = np.frombuffer(blob, dtype=np.uint8).copy()
        kind, = np.frombuffer(blob, dtype=np.uint32, count=1, offset=8)
        self.kind = int(kind)
        from_meta = self._meta()
        self.window, self.n_mel, self.n_out, self.enc_shape = from_meta

    def _meta(self):
        import struct
        blob = self._blob.tobytes()
        n = struct.unpack_from("<I", blob, 12)[0]
        meta = None
        want = b"crnn.meta" if self.kind == 1 else b"wave.meta"
        for i in range(n):
            base = 16 + 32 * i
            if blob[base:base + 24].split(b"\0", 1)[0] == want:
                off, cnt = struct.unpack_from("<II", blob, base + 24)
                meta = np.frombuffer(blob, np.int32, cnt, off)
        if self.kind == 1:
            return int(meta[1]), int(meta[0]), int(meta[12]), (2 * int(meta[11]),)
        return int(meta[0]), int(meta[1]), int(meta[5]), (int(meta[0]), int(meta[3]))

    # -- front end -------------------------------------------------------------
    def logmel(self, pcm: np.ndarray, divisor: float = 32767.0, clip: bool = True, preemph: float = 0.0,
               hop: int = 160, prefix: Optional[np.ndarray] = None) -> np.ndarray:
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        npre = 0 if prefix is None else int(prefix.size)
        pre = None if prefix is None else np.ascontiguousarray(prefix, dtype=np.float32)
        nf = int(lib().wwo_num_frames(pcm.size + npre, hop))
        mel = np.empty((nf, self.n_mel), np.float32)
        got = C.c_int64(0)
        _chk(lib().wwo_logmel(_p(self._blob), self._blob.size, _p(pcm), pcm.size, divisor, int(clip), preemph,
                              hop, _p(pre), npre, _p(mel), C.addressof(got)), "logmel")
        assert got.value == nf
        return mel

    def logmel_f32(self, x: np.ndarray, preemph: float = 0.0, hop: int = 160) -> np.ndarray:
        x = np.ascontiguousarray(x, dtype=np.float32)
        nf = int(lib().wwo_num_frames(x.size, hop))
        mel = np.empty((nf, self.n_mel), np.float32)
        got = C.c_int64(0)
        _chk(lib().wwo_logmel_f32(_p(self._blob), self._blob.size, _p(x), x.size, preemph, hop, _p(mel),
                                  C.addressof(got)), "logmel_f32")
        return mel

    # -- models ------------------------------------------------------------------
    def forward(self, windows: np.ndarray, want_enc: bool = False):
        w = np.ascontiguousarray(windows, dtype=np.float32)
        if w.ndim == 2:
            w = w[None]
        assert w.shape[1:] == (self.window, self.n_mel), w.shape
        B = w.shape[0]
        out = np.empty((B, self.n_out), np.float32)
        enc = np.empty((B,) + self.enc_shape, np.float32) if want_enc else None
        fn = lib().wwo_crnn_forward if self.kind == 1 else lib().wwo_wavenet_forward
        _chk(fn(_p(self._blob), self._blob.size, _p(w), B, _p(out), _p(enc)), "forward")
        return (out, enc) if want_enc else out

    def slide_forward(self, mel: np.ndarray, hop: int) -> np.ndarray:
        mel = np.ascontiguousarray(mel, dtype=np.float32)
        T = mel.shape[0]
        nw = (T - self.window) // hop + 1 if T >= self.window else 0
        out = np.empty((nw, self.n_out), np.float32)
        got = C.c_int64(0)
        _chk(lib().wwo_slide_forward(_p(self._blob), self._blob.size, _p(mel), T, hop, _p(out),
                                     C.addressof(got)), "slide_forward")
        assert got.value == nw
        return out


def stft_mag(frames: np.ndarray) -> np.ndarray:
    f = np.ascontiguousarray(frames, dtype=np.float32).reshape(-1, 512)
    mag = np.empty((f.shape[0], 257), np.float32)
    _chk(lib().wwo_stft_mag(_p(f), f.shape[0], _p(mag)), "stft_mag")
    return mag


def smooth(p: np.ndarray, w: int = 30) -> np.ndarray:
    p = np.ascontiguousarray(p, dtype=np.float64)
    out = np.empty_like(p)
    _chk(lib().wwo_smooth(_p(p), p.size, w, _p(out)), "smooth")
    return out


def far_frr(pos: np.ndarray, neg_smoothed: np.ndarray, thresholds: np.ndarray, num_wakewords: float,
            hours: float) -> Tuple[np.ndarray, np.ndarray, np.ndarray]:
    pos = np.ascontiguousarray(pos, dtype=np.float64)
    neg = np.ascontiguousarray(neg_smoothed, dtype=np.float64)
    thr = np.ascontiguousarray(thresholds, dtype=np.float64)
    frr = np.empty(thr.size, np.float64)
    fa = np.empty(thr.size, np.float64)
    cnt = np.empty(thr.size, np.int64)
    _chk(lib().wwo_far_frr(_p(pos), pos.size, _p(neg), neg.size, _p(thr), thr.size, num_wakewords, hours,
                           _p(frr), _p(fa), _p(cnt)), "far_frr")
    return frr, fa, cnt
