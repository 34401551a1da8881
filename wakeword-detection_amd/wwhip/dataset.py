"""Dataset -> H5 feature files, with the surface of ``utils/filter_dataset_to_h5.py``.

The reference walks a Hey-Snips style metadata JSON (``[{audio_file_path, is_hotword,
worker_id}, ...]``), pushes every wav through its never-reset ``Filter`` in 20 ms chunks
(``:63-112``), asks webrtcvad for the first / last speech chunk, and writes one float32
``[T, 40]`` dataset per clip with four attributes (``:136-145``).

Here the per-chunk Python loop becomes one closed-form frame schedule
(:func:`wwhip.evaluate.frame_schedule`, carry-over quirk C2 included) and ONE front-end launch
over all clips; the file is written by :mod:`wwhip.h5min` (h5py if it is installed).  webrtcvad
is a third-party C library that is not available here: the VAD is a plug-in
``vad(frame_bytes, sample_rate) -> bool`` with the reference's call signature; without one the
timestamps stay at the reference's "no speech found" value -1.
"""
from __future__ import annotations

import json
import os
from typing import Any, Callable, Dict, List, Optional, Sequence

import logging

import numpy as np

from .engine import Engine, frontend_params
from .evaluate import frame_schedule, read_wav
from .models import engine_for

_LOG = logging.getLogger(__name__)


def speech_bounds(samples: np.ndarray, frame_len: int, hop_len: int, sample_rate: int,
                  vad: Optional[Callable[[bytes, int], bool]]) -> tuple:
    """First / last speech position in hop units (``filter_dataset_to_h5.py:75-98``)."""
    start_ts, end_ts = -1, -1
    if vad is None:
        return start_ts, end_ts
    for start_idx in range(0, len(samples), frame_len):
        frame = samples[start_idx:start_idx + frame_len]
        if len(frame) < frame_len:
            frame = np.pad(frame, (0, frame_len - len(frame)), mode="constant")
        is_speech = bool(vad(np.int16(frame * 32768).tobytes(), sample_rate))
        if start_ts == -1 and is_speech:
            start_ts = start_idx // hop_len
        if start_ts > -1 and is_speech:
            end_ts = (start_idx + frame_len) // hop_len
    return start_ts, end_ts


def filter_clips(engine: Engine, clips: Sequence[np.ndarray], frame_len: int = 320, hop_len: int = 160,
                 carry_over: bool = True) -> List[np.ndarray]:
    """Log-mel features of every clip (float samples in [-1, 1)) as the reference's chunked
    ``Filter`` produces them: each clip is zero-padded to whole chunks; with ``carry_over`` the
    sample ring runs on across clips, so clip i > 0 also emits the frames that straddle the
    boundary with clip i-1 (quirk C2)."""
    padded = []
    for x in clips:
        x = np.asarray(x, dtype=np.float32)
        if len(x) % frame_len:
            x = np.pad(x, (0, frame_len - len(x) % frame_len), mode="constant")
        padded.append(x)
    if not padded:
        return []
    per_file, _ = frame_schedule([len(x) for x in padded], hop_len, frame_len, carry_over)
    fp = frontend_params(1.0, False, 0.0, hop_len, True)
    if not carry_over:
        return engine.logmel(padded, fp)
    mel = engine.logmel([np.concatenate(padded)], fp)[0]
    out, cur = [], 0
    for fpc in per_file:
        n = int(fpc.sum())
        out.append(mel[cur:cur + n])
        cur += n
    return out


class DatasetFilter:
    """``Dataset_Filter`` (``filter_dataset_to_h5.py:19-145``) without the per-chunk loop."""

    def __init__(self, dataset: str, models_dir: str, data_dir: str, out_dir: str, sample_rate: int = 16000,
                 frame_width: int = 20, hop_width: int = 10, vad: Optional[Callable[[bytes, int], bool]] = None,
                 loader: Optional[Callable[[str], np.ndarray]] = None, device: int = 0, **kwargs: Any) -> None:
        self.dataset = dataset
        with open(dataset, "r") as f:
            self.audio_metadata = json.load(f)
        self.speakers_dict = self.map_speakers()
        self.sr, self.fw, self.hw = sample_rate, frame_width, hop_width
        self.frame_len = self.sr // 1000 * self.fw
        self.hop_len = self.sr // 1000 * self.hw
        self.engine = engine_for(models_dir, device)
        self.num_filter_outputs = self.engine.n_mel
        self.data_dir, self.out_dir = data_dir, out_dir
        os.makedirs(out_dir, exist_ok=True)
        self.dataset_file = os.path.join(out_dir, os.path.basename(dataset).replace(".json", ".h5"))
        self.vad = vad
        self.load = loader or (lambda p: read_wav(p, self.sr))

    def map_speakers(self) -> Dict[Any, int]:
        speakers = []
        for d in self.audio_metadata:  # first-seen order (the reference iterates a set: arbitrary order)
            if d["worker_id"] not in speakers:
                speakers.append(d["worker_id"])
        return {s: i for i, s in enumerate(speakers)}

    def filter_dataset_audio(self) -> List[dict]:
        meta, clips = [], []
        for audio in self.audio_metadata:
            samples = np.asarray(self.load(os.path.join(self.data_dir, audio["audio_file_path"])), dtype=np.float32)
            if len(samples) == 0:  # the reference skips empty wavs before they reach the filter
                continue
            meta.append(audio)
            clips.append(samples)
        feats = filter_clips(self.engine, clips, self.frame_len, self.hop_len, carry_over=True)
        audio_clips = []
        for audio, samples, f in zip(meta, clips, feats):
            if len(f) == 0:
                continue
            s_ts, e_ts = speech_bounds(samples, self.frame_len, self.hop_len, self.sr, self.vad)
            audio_clips.append({
                "file_name": os.path.basename(audio["audio_file_path"]).replace(".wav", ""),
                "is_hotword": audio["is_hotword"],
                "features": f,
                "speech_start_ts": s_ts,
                "speech_end_ts": e_ts,
                "speaker": self.speakers_dict[audio["worker_id"]],
            })
        self.write_h5(audio_clips)
        return audio_clips

    def write_h5(self, audio_clips: Sequence[dict]) -> None:
        write_h5(self.dataset_file, audio_clips)


def write_h5(path: str, audio_clips: Sequence[dict]) -> None:
    """One dataset per clip + the four attributes (``filter_dataset_to_h5.py:136-145``)."""
    try:
        import h5py  # type: ignore
    except ImportError:
        from . import h5min
        # the built-in writer emits the same "earliest"-format structures h5py does: real libhdf5 reads its files back
        # (tests/test_h5min.py::test_h5min_files_open_with_libhdf5, h5py 3.3.0 / libhdf5 1.10.6 of the image's conda python) and
        # the built-in reader reads h5py-written ones bit for bit (::test_h5min_reads_a_file_written_by_h5py)
        _LOG.info("h5py is not installed: writing %s with the built-in HDF5 writer (wwhip.h5min)", path)
        h5min.write_datasets(path, {
            c["file_name"]: (np.asarray(c["features"], np.float32),
                             {"is_hotword": c["is_hotword"], "speaker": c["speaker"],
                              "speech_start_ts": c["speech_start_ts"], "speech_end_ts": c["speech_end_ts"]})
            for c in audio_clips})
        return
    _LOG.info("writing %s with h5py", path)
    with h5py.File(path, "w") as h5f:  # pragma: no cover - h5py is not installed in this image
        for c in audio_clips:
            dset = h5f.create_dataset(c["file_name"], data=np.asarray(c["features"], np.float32))
            for k in ("is_hotword", "speaker", "speech_start_ts", "speech_end_ts"):
                dset.attrs[k] = c[k]
