"""Offline evaluation with the surface of the reference's evaluators.

* :func:`get_posterior` - ``utils/evaluate_models.py:26-108``: per wav, pad 0.5 s of zeros on
  both sides, 20 ms chunks through the (never reset) ``Filter``, one inference per chunk once
  ``encoder_len`` mel frames are buffered, hop 2 frames; max per positive clip / all windows of
  the negative stream.
* :func:`far_frr` - numeric core of ``plot_FRR_FAR`` (``:183-218``): 30-tap moving average of
  the negative stream, FRR and false-accepts/hour per threshold (rising edges).
* :func:`load_data` / :func:`models_predict` - ``utils/evaluate_tf_lite_opts.py:35-69``: one
  zero-padded window per H5 clip, class = posterior >= 0.5.

The per-sample / per-chunk Python loops of the reference are replaced by closed-form frame and
window schedules plus three batched GPU launches (front end, windows, sweep); the schedule
below reproduces the reference's chunking quirks exactly (frames credited to a file, the
"one inference per chunk" rule, windows dropped at end of file).
"""
from __future__ import annotations

import wave
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .engine import Engine, frontend_params
from .models import engine_for

WINDOW = 512


def read_wav(path: str, sample_rate: int = 16000) -> np.ndarray:
    """PCM16 mono wav -> float32 in [-1, 1) exactly as ``librosa.load(path, sr=sample_rate)``
    returns it when no resampling is needed (int16 / 32768)."""
    with wave.open(path, "rb") as w:
        if w.getsampwidth() != 2:
            raise ValueError(f"{path}: only 16-bit PCM is supported")
        if w.getframerate() != sample_rate:
            raise ValueError(f"{path}: sample rate {w.getframerate()} != {sample_rate} (no resampler here)")
        raw = np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
        if w.getnchannels() > 1:  # librosa averages channels (mono=True)
            raw = raw.reshape(-1, w.getnchannels()).astype(np.float32).mean(axis=1) / np.float32(32768.0)
            return raw.astype(np.float32)
    return raw.astype(np.float32) / np.float32(32768.0)


def frame_schedule(n_padded_per_file: Sequence[int], hop: int = 160, chunk: int = 320,
                   carry_over: bool = True) -> Tuple[List[np.ndarray], List[int]]:
    """For each file (length already padded to whole chunks): number of mel frames produced by
    every chunk, and the stream position (in samples) at which the file starts.  With
    ``carry_over`` the sample ring is continuous across files (reference behaviour, quirk C2)."""
    per_file, starts = [], []
    pos = 0  # samples pushed into the ring so far
    for n in n_padded_per_file:
        if not carry_over:
            pos = 0
        starts.append(pos)
        ends = pos + chunk * np.arange(1, n // chunk + 1)
        tot = np.where(ends >= WINDOW, (ends - WINDOW) // hop + 1, 0)
        prev = (pos - WINDOW) // hop + 1 if pos >= WINDOW else 0
        per_file.append(np.diff(np.concatenate(([prev], tot))))
        pos += n
    return per_file, starts


def window_schedule(frames_per_chunk: np.ndarray, encoder_len: int, hop: int = 2) -> np.ndarray:
    """Start row (within the file's frame list) of every window the reference evaluates:
    at most one inference per chunk, taken when >= encoder_len frames are buffered, after which
    ``hop`` frames are dropped (evaluate_models.py:66-73)."""
    starts = []
    have = 0  # frames buffered
    base = 0  # index of the first buffered frame
    for k in frames_per_chunk:
        have += int(k)
        if have >= encoder_len:
            starts.append(base)
            base += hop
            have -= hop
    return np.array(starts, dtype=np.int64)


def get_posterior(models_dir, model_type, eval_type, test_files, frame_width, sample_rate, examine_audio=False,
                  loader: Optional[Callable[[str], np.ndarray]] = None, carry_over: bool = True,
                  device: int = 0) -> list:
    """Reference signature (``utils/evaluate_models.py:26-28``) plus ``loader`` (path -> float
    samples; default :func:`read_wav`) and ``carry_over`` (False = reset the ring per file)."""
    if model_type not in ("CRNN", "Wavenet"):
        raise ValueError("model_type must be 'CRNN' or 'Wavenet'")
    eng: Engine = engine_for(models_dir, device)
    encoder_len = eng.window
    frame_length = sample_rate // 1000 * frame_width
    hop = 2
    load = loader or (lambda p: read_wav(p, sample_rate))
    pidx = eng.posterior_index
    pad = sample_rate // 2
    # the ring is continuous over all files: build one stream, remember the file boundaries
    signals = []
    for f in test_files:
        x = np.asarray(load(f), dtype=np.float32)
        x = np.pad(x, (pad, pad), mode="constant")
        if len(x) % frame_length:
            x = np.pad(x, (0, frame_length - len(x) % frame_length), mode="constant")
        signals.append(x)
    if not signals:
        return []
    per_file, starts = frame_schedule([len(s) for s in signals], 160, frame_length, carry_over)
    if carry_over:
        stream = [np.concatenate(signals)]
    else:
        stream = signals
    mels = eng.logmel(stream, frontend_params(1.0, False, 0.0, 160, True))
    all_posterior: list = []
    frame_cursor = 0
    for i, fpc in enumerate(per_file):
        n_frames = int(fpc.sum())
        if carry_over:
            mel = mels[0][frame_cursor:frame_cursor + n_frames]
            frame_cursor += n_frames
        else:
            mel = mels[i]
        ws = window_schedule(fpc, encoder_len, hop)
        if len(ws):
            # windows start every `hop` rows from row 0 -> one sliding launch
            post = eng.slide_forward(mel[: ws[-1] + encoder_len], hop)[:, pidx]
        else:
            post = np.zeros(0, np.float32)
        if eval_type == "false_negatives":
            all_posterior.append(np.max(post))  # raises on an empty clip, like the reference
        else:
            all_posterior.extend(post.tolist())
    return all_posterior


def default_thresholds() -> np.ndarray:
    return np.arange(0.5, 0.99999, 0.005)  # evaluate_models.py:185


def far_frr(keyword_posteriors, no_keyword_posteriors, num_wakewords, total_duration_hrs, thresholds=None,
            windowsize: int = 30, engine: Optional[Engine] = None, models_dir: Optional[str] = None):
    """Returns ``(thresholds, FRR, FA_per_hour, FA_count)`` as ``plot_FRR_FAR`` computes them."""
    eng = engine or engine_for(models_dir)
    thr = default_thresholds() if thresholds is None else np.asarray(thresholds, np.float64)
    frr, fa, cnt = eng.far_frr(np.asarray(keyword_posteriors, np.float32), np.asarray(no_keyword_posteriors, np.float32),
                               thr, float(num_wakewords), float(total_duration_hrs), window=windowsize)
    return thr, frr, fa, cnt


def frr_at_fa(frr: np.ndarray, far: np.ndarray, fa_limit: float = 0.5) -> float:
    """BASELINE metric: min FRR over the thresholds whose FA/h <= ``fa_limit``."""
    ok = np.asarray(far) <= fa_limit
    return float(np.min(np.asarray(frr)[ok])) if ok.any() else float("nan")


def load_data(features: Sequence[np.ndarray], labels: Sequence[int], timesteps: int, num_features: int):
    """Truncate each clip's ``[T, F]`` features to ``timesteps`` rows and zero-pad at the end
    (``evaluate_tf_lite_opts.py:35-47``).  Takes arrays instead of an H5 path (h5py is optional:
    see :func:`load_h5`)."""
    X = np.zeros((len(features), timesteps, num_features), dtype=np.float32)
    for i, f in enumerate(features):
        f = np.asarray(f, dtype=np.float32)[:timesteps]
        X[i, : f.shape[0], : f.shape[1]] = f
    return X, np.array(labels, dtype=np.uint8)


def load_h5(data_file: str, timesteps: int, num_features: int):
    """The reference's on-disk format (one dataset per clip, attr ``is_hotword``,
    ``filter_dataset_to_h5.py:136-145``); needs h5py at run time."""
    try:
        import h5py  # type: ignore
    except ImportError as e:  # pragma: no cover
        raise RuntimeError("reading .h5 feature files needs h5py, which is not installed here") from e
    feats, labels = [], []
    with h5py.File(data_file, "r") as h5:
        for key in h5.keys():
            labels.append(h5[key].attrs["is_hotword"])
            feats.append(h5[key][()])
    return load_data(feats, labels, timesteps, num_features)


def models_predict(engine: Engine, X: np.ndarray, threshold: float = 0.5) -> Tuple[List[int], np.ndarray]:
    """One window per clip, class 1 when posterior >= threshold (``evaluate_tf_lite_opts.py:49-69``)."""
    post = engine.forward(X)[:, engine.posterior_index]
    return [1 if p >= threshold else 0 for p in post], post
