"""ORACLE (test infrastructure only) - NumPy restatement of the reference's host-side
front end and evaluator, written to follow the reference statement by statement.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may
import this.  The pieces restated here are NumPy code in the reference and run with the
same NumPy calls (``np.hanning``, ``np.fft.rfft``, ``np.abs``, ``np.convolve``), so they
are as close to "the reference run here" as the missing TFLite runtime allows; the
framing grid is additionally pinned by fixtures produced by importing the reference's
``RingBuffer`` (``tests/golden/make_golden.py``).  The mel/encode/detect graphs are
evaluated by ``oracle/tflite_interp.py`` ("parity unpinned", see its header).
"""
from __future__ import annotations

from typing import Callable, List, Optional, Sequence

import numpy as np


class RefRing:
    """Index arithmetic of reference ``spokestack/ring_buffer.py:9-130`` (one spare slot)."""

    def __init__(self, capacity: int, item_shape=(), dtype=np.float32) -> None:
        self._n = capacity + 1  # ring_buffer.py:15
        self._buf = np.empty((self._n,) + tuple(item_shape), dtype=dtype)
        self._r = 0
        self._w = 0

    @property
    def is_empty(self) -> bool:
        return self._r == self._w  # :28

    @property
    def is_full(self) -> bool:
        return self._r == (self._w + 1) % self._n  # :37

    def rewind(self):
        self._r = (self._w + 1) % self._n  # :54
        return self

    def seek(self, k: int):
        self._r = (self._r + k) % self._n  # :88
        return self

    def reset(self):
        self._w = self._r  # :63
        return self

    def fill(self, v):
        self._buf.fill(v)  # :75-76
        self._r = (self._w + 1) % self._n
        return self

    def write(self, item) -> None:
        if self.is_full:
            raise IndexError("Buffer is full")  # :100-101
        self._buf[self._w] = item
        self._w = (self._w + 1) % self._n

    def read(self):
        if self.is_empty:
            raise IndexError("Buffer is empty")  # :112-113
        item = self._buf[self._r : self._r + 1]
        self._r = (self._r + 1) % self._n
        return item

    def read_all(self):
        self.rewind()  # :126
        cur = []
        while not self.is_empty:
            cur.append(self.read())
        return np.concatenate(cur).astype(self._buf.dtype)


class RefFilter:
    """Reference ``utils/tf_lite/filter.py:9-79`` with ``filter_model`` supplied as a callable
    ``[1,257] f32 -> [1,40] f32`` (the TFLite graph evaluated by the op-by-op oracle)."""

    def __init__(self, filter_model: Callable[[np.ndarray], np.ndarray], pre_emphasis: float = 0.0,
                 sample_rate: int = 16000, fft_hop_length: int = 10, window_size: int = 512) -> None:
        self.pre_emphasis = pre_emphasis
        self.hop_length = int(fft_hop_length * sample_rate / 1000)  # filter.py:19
        self.filter_model = filter_model
        self._window_size = window_size  # filter.py:31
        self._fft_window = np.hanning(self._window_size)  # filter.py:32
        self.sample_window = RefRing(self._window_size)  # filter.py:35
        self._prev_sample = 0.0

    def filter_frame(self, frame: np.ndarray) -> List[np.ndarray]:
        prev_sample = frame[-1]  # filter.py:42
        frame -= self.pre_emphasis * np.append(self._prev_sample, frame[:-1])  # :43 (in place)
        self._prev_sample = prev_sample
        feats = []
        for sample in frame:  # :50
            self.sample_window.write(sample)
            if self.sample_window.is_full:
                feats.append(self._analyze().squeeze())
                self.sample_window.rewind().seek(self.hop_length)  # :55
        return feats

    def stft_mag(self) -> np.ndarray:
        fr = self.sample_window.read_all()  # :62
        fr = np.fft.rfft(fr * self._fft_window, n=self._window_size)  # :63 (float64)
        return np.abs(fr).astype(np.float32)  # :64

    def _analyze(self) -> np.ndarray:
        return self.filter_model(np.expand_dims(self.stft_mag(), 0))  # :72-73


def normalise_pcm(frame_i16: np.ndarray) -> np.ndarray:
    """Reference ``spokestack/wakeword/tflite.py:150-151``."""
    f = frame_i16.astype(np.float32) / (2 ** 15 - 1)
    return np.clip(f, -1.0, 1.0)


def sliding_posteriors(filt: RefFilter, samples: np.ndarray, encoder_len: int,
                       window_fn: Callable[[np.ndarray], float], frame_length: int = 320,
                       inference_hop: int = 2, pad: int = 8000) -> List[float]:
    """Per-file loop of reference ``utils/evaluate_models.py:45-88``.

    ``samples`` are the floats librosa would return; ``window_fn`` maps a ``[T,40]`` window to
    the scalar posterior.  Note the reference runs at most ONE inference per 20 ms chunk
    (``if``, not ``while``, at :70) and discards the windows still pending at end of file."""
    out: List[float] = []
    window_buffer: List[np.ndarray] = []
    samples = np.pad(samples, (pad, pad), mode="constant")  # :52-53
    for start in np.arange(0, len(samples), frame_length):  # :57
        frame = samples[start : start + frame_length]
        if len(frame) < frame_length:
            frame = np.pad(frame, (0, frame_length - len(frame)), mode="constant")  # :59-61
        feats = filt.filter_frame(frame)  # :64
        if len(feats) > 0:
            window_buffer.extend(feats)
        if len(window_buffer) >= encoder_len:  # :70
            frames = window_buffer[:encoder_len]
            window_buffer = window_buffer[inference_hop:]
            out.append(window_fn(np.array(frames)))
    return out


def far_frr(keyword_post: np.ndarray, no_keyword_post: np.ndarray, num_wakewords: int, hours: float,
            thresholds: Optional[np.ndarray] = None, windowsize: int = 30):
    """Reference ``utils/evaluate_models.py:183-218`` (numbers only, no plots)."""
    if thresholds is None:
        thresholds = np.arange(0.5, 0.99999, 0.005)  # :185
    neg = np.convolve(no_keyword_post, np.ones((windowsize,)) / windowsize, mode="same")  # :188-189
    frr, far, cnt = [], [], []
    for threshold in thresholds:
        accepts = (keyword_post > threshold).sum()  # :201
        frr.append((num_wakewords - accepts) / num_wakewords)  # :202-204
        prev = False
        fa = 0
        for p in neg:  # :211-216
            if p > threshold and not prev:
                fa += 1
            prev = bool(p > threshold)
        cnt.append(fa)
        far.append(fa / hours)  # :217
    return np.array(frr), np.array(far), np.array(cnt), neg


def frr_at_fa(frr: np.ndarray, far: np.ndarray, fa_limit: float = 0.5) -> float:
    """BASELINE.json metric 'FRR @ 0.5 FA/h': min FRR over thresholds with FA/h <= limit
    (SURVEY 8d)."""
    ok = far <= fa_limit
    return float(np.min(frr[ok])) if ok.any() else float("nan")


def load_h5_like(features: Sequence[np.ndarray], timesteps: int, num_features: int) -> np.ndarray:
    """Reference ``utils/evaluate_tf_lite_opts.py:35-47``: truncate to ``timesteps`` rows and
    zero-pad at the end."""
    X = np.zeros((len(features), timesteps, num_features), dtype=np.float32)
    for i, f in enumerate(features):
        f = np.asarray(f)[:timesteps]
        X[i, : f.shape[0], : f.shape[1]] = f
    return X


def wfst_smooth(posterior_probs) -> List[int]:
    """Reference ``wwdetect/wfst.py:17-71`` without pynini: the single shortest path through the
    2-state lattice it builds, in OpenFst's arithmetic (TropicalWeight = float32; distances relaxed
    in state order, replaced only when strictly smaller, so ties keep state 0).  Returns the state
    sequence (0 = 'other', 1 = 'wakeword')."""
    with np.errstate(divide="ignore"):
        obs = -np.log(np.asarray(posterior_probs, dtype=np.float32))  # wfst.py:33 (float32 array)
    T, P = obs.shape
    d = [np.float32(-np.log(1.0 / P) + obs[0, p]) for p in range(P)]  # :56-57, Arc weight -> float32
    back = np.zeros((T, P), np.int64)
    for t in range(1, T):  # :59-66
        nd = []
        for p_to in range(P):
            best, arg = None, 0
            for p_from in range(P):
                cost = obs[t, p_to]
                if p_to == p_from:
                    cost = np.float32(cost - 1)  # test_transition_matrix == 1 everywhere
                cand = np.float32(d[p_from] + cost)
                if best is None or cand < best:
                    best, arg = cand, p_from
            nd.append(best)
            back[t, p_to] = arg
        d = nd
    st = 1 if d[1] < d[0] else 0
    path = [0] * T
    for t in range(T - 1, -1, -1):
        path[t] = st
        st = int(back[t, st])
    return path


class RefKeywordRecognizer:
    """Reference ``spokestack/asr/keyword/tflite.py:15-195`` on ``RefRing``s, with the three models supplied as
    callables (``filter_model``: ``[1,257] -> [1,40]``; ``encode_model(window, state) -> (frame, state)``;
    ``detect_model(window) -> [[posteriors]]``).  ``context`` needs ``is_active``, ``transcript``, ``confidence``, ``event``."""

    def __init__(self, classes, filter_model, encode_model, detect_model, mel_length: int, mel_width: int, state_shape,
                 encode_length: int, encode_width: int, pre_emphasis: float = 0.97, sample_rate: int = 16000,
                 fft_hop_length: int = 10, posterior_threshold: float = 0.5, window_size: int = 512) -> None:
        self.classes = classes
        self.pre_emphasis = pre_emphasis
        self.hop_length = int(fft_hop_length * sample_rate / 1000)  # :41
        self.filter_model, self.encode_model, self.detect_model = filter_model, encode_model, detect_model
        self._window_size = window_size  # :62
        self._fft_window = np.hanning(self._window_size)  # :63
        self.state = np.zeros(state_shape, np.float32)  # :75
        self.sample_window = RefRing(self._window_size)  # :79-81
        self.frame_window = RefRing(mel_length, (mel_width,))  # :82-84
        self.encode_window = RefRing(encode_length, (encode_width,))  # :85-87
        self.frame_window.fill(0.0)  # :91
        self.encode_window.fill(-1.0)  # :92
        self._posterior_threshold = posterior_threshold
        self._prev_sample = 0.0
        self._is_active = False

    def __call__(self, context, frame) -> None:  # :98-105
        self._sample(context, frame)
        if not context.is_active and self._is_active:
            self._detect(context)
        self._is_active = context.is_active

    def _sample(self, context, frame) -> None:  # :107-129
        frame = normalise_pcm(frame)
        prev_sample = frame[-1]
        frame -= self.pre_emphasis * np.append(self._prev_sample, frame[:-1])
        self._prev_sample = prev_sample
        for sample in frame:
            self.sample_window.write(sample)
            if self.sample_window.is_full:
                if context.is_active:
                    self._analyze()
                self.sample_window.rewind().seek(self.hop_length)

    def _analyze(self) -> None:  # :131-158
        fr = self.sample_window.read_all()
        fr = np.abs(np.fft.rfft(fr * self._fft_window, n=self._window_size)).astype(np.float32)
        mel = self.filter_model(np.expand_dims(fr, 0))[0]
        self.frame_window.rewind().seek(1)
        self.frame_window.write(mel)
        win = np.expand_dims(self.frame_window.read_all(), 0)
        enc, self.state = self.encode_model(win, self.state)
        self.encode_window.rewind().seek(1)
        self.encode_window.write(enc)

    def _detect(self, context) -> None:  # :165-184
        win = np.expand_dims(self.encode_window.read_all(), 0)
        posterior = self.detect_model(win)[0][0]
        class_index = np.argmax(posterior)
        confidence = posterior[class_index]
        if confidence >= self._posterior_threshold:
            context.transcript = self.classes[class_index]
            context.confidence = confidence
            context.event("recognize")
        else:
            context.event("timeout")
        self.reset()

    def reset(self) -> None:  # :186-191
        self.sample_window.reset()
        self.frame_window.reset().fill(0.0)
        self.encode_window.reset().fill(-1.0)
        self.state[:] = 0.0


class RefGatedStream:
    """The gating of reference ``WakewordTrigger`` around its models, one stream (``spokestack/wakeword/tflite.py``):
    ``__call__`` :123-146 (VAD falling edge; an ACTIVE context is not sampled at all; ``reset()`` on the fall), ``_sample``
    :148-168 (normalise, clip, pre-emphasis with ``_prev_sample`` carried over - and NOT touched by a reset -, the sample ring
    always advances, a full ring is analysed only while ``is_speech``), ``_filter`` :181-191 (the mel window slides by one row per
    analysed frame), ``reset`` :241-246 (sample ring emptied, mel window zeroed).

    The models are left out: :meth:`tick` returns the mel WINDOWS ``[k][T][40]`` (k = 0, 1 or 2) the encoder would see in this
    tick, for the caller to evaluate in one batch; ``mel_row(frame512 float32) -> [40]`` is the STFT + filter graph (the C
    oracle's ``logmel_f32`` on one frame).  The sample ring is kept as an array and cut at 512 instead of written sample by
    sample - the same frames as the reference's per-sample loop (framing pinned by tests/golden/framing.npz)."""

    def __init__(self, mel_row: Callable[[np.ndarray], np.ndarray], mel_length: int, mel_width: int = 40,
                 pre_emphasis: float = 0.0, window: int = 512, hop: int = 160) -> None:
        self.mel_row, self.T, self.F = mel_row, mel_length, mel_width
        self.pre_emphasis, self.window, self.hop = pre_emphasis, window, hop
        self._is_speech = False
        self._prev_sample = np.float32(0.0)
        self.samples = np.zeros(0, np.float32)
        self.frames = np.zeros((self.T, self.F), np.float32)

    def tick(self, frame_i16: np.ndarray, is_speech: bool, is_active: bool) -> List[np.ndarray]:
        vad_fall = self._is_speech and not is_speech  # :134
        self._is_speech = is_speech
        out: List[np.ndarray] = []
        if not is_active:  # :139-140
            frame = normalise_pcm(frame_i16)  # :150-151
            prev_sample = frame[-1]  # :156
            frame = frame - np.float32(self.pre_emphasis) * np.append(self._prev_sample, frame[:-1]).astype(np.float32)  # :157
            self._prev_sample = prev_sample
            self.samples = np.concatenate([self.samples, frame.astype(np.float32)])
            while len(self.samples) >= self.window:  # :163-168
                if is_speech:
                    row = self.mel_row(self.samples[:self.window])
                    self.frames = np.concatenate([self.frames[1:], row.reshape(1, self.F)])  # :186-187
                    out.append(self.frames.copy())
                self.samples = self.samples[self.hop:]
        if vad_fall:  # :143-146
            self.reset()
        return out

    def reset(self) -> None:
        self.samples = np.zeros(0, np.float32)  # :243
        self.frames = np.zeros((self.T, self.F), np.float32)  # :244
