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

import os
import wave
from typing import Callable, List, Optional, Sequence, Tuple

import numpy as np

from .engine import Engine, frontend_params
from .models import engine_for

WINDOW = 512
CLIP_PAD = 8000  # 0.5 s of zeros each side of a clip (evaluate_models.py:52-53)
CLIP_HOP = 2     # mel rows between windows (evaluate_models.py:42)
_PIN = None  # page-locked staging buffer of clip_posteriors (torch tensor, grown on demand)
_COPY_THREADS = 8  # copy threads of the uploader that assembles a rank's samples in page-locked memory


class _Phases:
    """Wall-clock seconds per host phase of an evaluation call (``timing`` dict of the sharded flows): ``with ph("name"):``
    adds the block's duration to ``timing[name]``; a ``None`` dict makes it a no-op."""

    def __init__(self, timing: Optional[dict]) -> None:
        self.t = timing

    def __call__(self, name: str):
        import contextlib
        import time

        @contextlib.contextmanager
        def scope():
            if self.t is None:
                yield
                return
            t0 = time.perf_counter()
            try:
                yield
            finally:
                self.t[name] = self.t.get(name, 0.0) + time.perf_counter() - t0
        return scope()


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
    ``hop`` frames are dropped (evaluate_models.py:66-73).

    Closed form of that loop: with c_j the frames emitted up to and including chunk j, window n needs c_j >= T + hop n and a
    chunk of its own after window n - 1's, so it is taken at chunk j_n = max(m_n, j_{n-1} + 1) with m_n the first such j, i.e.
    j_n = n + max_{i <= n} (m_i - i); windows exist while j_n is a chunk of the file.  (The literal loop is
    ``_window_schedule_loop``; tests/test_host_logic.py holds the two together and against the reference's RingBuffer-driven loop.)"""
    fpc = np.asarray(frames_per_chunk, np.int64)
    if fpc.size == 0:
        return np.zeros(0, np.int64)
    c = np.cumsum(fpc)
    n_max = int((c[-1] - encoder_len) // hop + 1) if c[-1] >= encoder_len else 0
    if n_max <= 0:
        return np.zeros(0, np.int64)
    n = np.arange(n_max, dtype=np.int64)
    m = np.searchsorted(c, encoder_len + hop * n, side="left")
    j = n + np.maximum.accumulate(m - n)
    return hop * n[: int(np.count_nonzero(j < fpc.size))]


def _window_schedule_loop(frames_per_chunk: np.ndarray, encoder_len: int, hop: int = 2) -> np.ndarray:
    """The reference's loop, statement for statement (what :func:`window_schedule` is the closed form of)."""
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
    """Reference signature (``utils/evaluate_models.py:26-28``) plus ``loader`` (path -> samples: int16 PCM or float32 in
    [-1, 1); default :func:`read_wav_pcm`) and ``carry_over`` (False = reset the ring per file).  One GPU; the same
    implementation as :func:`get_posterior_sharded` at world size 1 (one front-end launch over all files, one model launch
    over all windows)."""
    return get_posterior_sharded(models_dir, model_type, eval_type, list(test_files), frame_width, sample_rate, 0, 1,
                                 loader=loader, carry_over=carry_over, device=device)


# ----------------------------------------------------------------------------------------------
# The same flow, sharded over ranks: utterance-sharded for the positives, ONE long
# negative stream cut into contiguous posterior ranges for the false accepts.
# ----------------------------------------------------------------------------------------------
def wav_length(path: str, sample_rate: int = 16000) -> int:
    """Samples ``read_wav(path)`` would return, from the header alone."""
    with wave.open(path, "rb") as w:
        if w.getframerate() != sample_rate:
            raise ValueError(f"{path}: sample rate {w.getframerate()} != {sample_rate} (no resampler here)")
        return w.getnframes()


def read_wav_pcm(path: str, sample_rate: int = 16000) -> np.ndarray:
    """Mono PCM16 wav as int16 (``librosa.load`` = these / 32768); anything else as :func:`read_wav`'s float32."""
    with wave.open(path, "rb") as w:
        if w.getsampwidth() == 2 and w.getnchannels() == 1 and w.getframerate() == sample_rate:
            return np.frombuffer(w.readframes(w.getnframes()), dtype=np.int16)
    return read_wav(path, sample_rate)


class StreamPlan:
    """Window layout of ``get_posterior`` over a list of files, as pure arithmetic on the file lengths - identical on
    every rank.  File ``k`` (padded by 0.5 s each side, then to whole chunks) starts at stream sample ``pos[k]``; the
    never-reset ring (quirk C2) credits it the global frames ``[F[k], F[k] + n_frames[k])`` (global frame ``j`` = stream
    samples ``[160 j, 160 j + 512)``; the first ones of a later file start in its predecessor's tail) and the
    one-inference-per-chunk rule (``evaluate_models.py:70``) gives it ``n_win[k]`` windows, window ``i`` covering the
    global frames ``[F[k] + hop i, F[k] + hop i + T)``."""

    def __init__(self, lengths: Sequence[int], encoder_len: int, frame_length: int = 320, sample_rate: int = 16000,
                 hop: int = 2, carry_over: bool = True) -> None:
        self.T, self.hop, self.frame_length, self.pad, self.carry = int(encoder_len), int(hop), int(frame_length), sample_rate // 2, carry_over
        self.lengths = np.asarray(lengths, np.int64)
        padded = self.lengths + 2 * self.pad
        self.padded = padded + (-padded) % frame_length
        if frame_length == 160 * self.hop:
            self._closed_form()
        else:
            self._per_file()
        self.offs = np.concatenate(([0], np.cumsum(self.n_win)))

    def _per_file(self) -> None:
        """The schedules file by file (:func:`frame_schedule`, :func:`window_schedule`): any chunk length."""
        per_file, starts = frame_schedule(self.padded.tolist(), 160, self.frame_length, self.carry)
        self.pos = np.asarray(starts, np.int64)
        self.n_frames = np.array([int(f.sum()) for f in per_file], np.int64)
        # first global frame credited to the file: the ring emits frames in order, so it is the count emitted before it
        self.F = np.array([((p - WINDOW) // 160 + 1 if p >= WINDOW else 0) for p in self.pos], np.int64)
        # (window i of a file starts at row hop * i of its frame list: the schedule drops `hop` rows per inference)
        self.n_win = np.array([len(window_schedule(fpc, self.T, self.hop)) for fpc in per_file], np.int64)

    def _closed_form(self) -> None:
        """The same numbers for ALL files at once when a chunk is exactly ``hop`` mel hops (the reference: 320 samples, hop 2),
        as arithmetic on the cumulative padded lengths - what a 2,500-file test split needs instead of 2,500 Python-level
        schedules.  With tot(p) = frames the ring has emitted once p samples went in: a file at stream position pos has
        F = tot(pos) frames before it and n_frames = tot(pos + padded) - F of its own; after its chunk j it holds
        c_j = tot(pos + C (j + 1)) - F, so window n (needs c_j >= T + hop n) is ready at chunk
        m_n = max(0, ceil(A / C) - 1 + n) with A = 512 + 160 (F + T - 1) - pos and C = 160 hop: m_n - n is constant, and
        :func:`window_schedule`'s j_n = n + max_{i <= n} (m_i - i) = n + max(ceil(A / C) - 1, 0).  Windows exist while j_n is a
        chunk of the file and n < n_max.  (tests/test_host_logic.py holds this against the per-file schedules and those
        against the reference's loop.)"""
        C = self.frame_length
        ends = np.cumsum(self.padded)
        self.pos = (ends - self.padded) if self.carry else np.zeros_like(self.padded)
        tot = lambda p: np.where(p >= WINDOW, (p - WINDOW) // 160 + 1, 0)  # noqa: E731
        self.F = tot(self.pos)
        self.n_frames = tot(self.pos + self.padded) - self.F
        J = self.padded // C
        A = WINDOW + 160 * (self.F + self.T - 1) - self.pos
        e0 = np.maximum(-((-A) // C) - 1, 0)
        n_max = np.where(self.n_frames >= self.T, (self.n_frames - self.T) // self.hop + 1, 0)
        self.n_win = np.clip(np.minimum(n_max, J - e0), 0, None).astype(np.int64)

    @property
    def total(self) -> int:
        return int(self.offs[-1])

    def shares(self, eval_type: str, world: int) -> List[List[Tuple[int, int, int]]]:
        """Per rank: ``(file, i0, i1)`` runs of windows.  Positives: whole files dealt longest-first round-robin
        (``dist.shard_by_length``); the negative stream: contiguous posterior ranges (``dist.split_stream``), cut
        where a range crosses a file boundary."""
        return [[tuple(r) for r in a.tolist()] for a in self.shares_arr(eval_type, world)]

    def shares_arr(self, eval_type: str, world: int, ranks: Optional[Sequence[int]] = None) -> List[np.ndarray]:
        """:meth:`shares` as one ``int64 [n, 3]`` array per rank, built without a Python-level pass over the files (every
        rank works out EVERY rank's share - the gather needs to know where any rank's values go - so at 5,000 files and 8
        ranks the per-file loops were the part of a call that no rank count divides).  ``ranks``: only these ranks' shares, in
        this order (a job needs its own share to start and the others only when the gather comes: :class:`_PosteriorJob`)."""
        from . import dist as D
        out: List[np.ndarray] = []
        which = range(world) if ranks is None else ranks
        if eval_type == "false_negatives":
            order = getattr(self, "_order", None)
            if order is None:
                order = self._order = np.argsort(-self.lengths, kind="stable")  # = dist.shard_by_length: rank r gets order[r::world]
            for r in which:
                files = np.sort(order[r::world])
                files = files[self.n_win[files] > 0]
                out.append(np.stack((files, np.zeros_like(files), self.n_win[files]), axis=1) if len(files) else np.zeros((0, 3), np.int64))
        else:
            offs = self.offs
            ranges = D.split_stream(self.total, world)
            for lo, hi in (ranges[r] for r in which):
                if hi <= lo:
                    out.append(np.zeros((0, 3), np.int64))
                    continue
                k0 = int(np.searchsorted(offs, lo, side="right")) - 1
                k1 = int(np.searchsorted(offs, hi, side="left"))
                ks = np.arange(k0, k1, dtype=np.int64)
                i0 = np.maximum(lo, offs[ks]) - offs[ks]
                i1 = np.minimum(hi, offs[ks + 1]) - offs[ks]
                ok = i1 > i0
                out.append(np.stack((ks[ok], i0[ok], i1[ok]), axis=1))
        return out

    def sample_range(self, k: int, i0: int, i1: int) -> Tuple[int, int]:
        """Stream samples the windows ``i0..i1-1`` of file ``k`` are functions of (with pre-emphasis 0, as ``Filter``'s
        default: no sample before the first frame is needed)."""
        g0 = int(self.F[k]) + self.hop * i0
        g1 = int(self.F[k]) + self.hop * (i1 - 1) + self.T - 1
        return 160 * g0, 160 * g1 + WINDOW  # (without carry every file has its own stream: pos = 0, F = 0)


class JoinedPCM:
    """``concatenate_FA``'s one long wav (``evaluate_models.py:150-160``) WITHOUT materialising it: the clips stay where they
    are and ``gap`` zero samples lie between them; the staging code copies each clip straight into the upload buffer (one
    pass over two hours of audio instead of three).  ``len()``, ``dtype`` and ``to_array()`` are all a caller needs."""

    dtype = np.dtype(np.int16)

    def __init__(self, clips: Sequence[np.ndarray], gap: int, lengths: Optional[np.ndarray] = None) -> None:
        # the clips are LOOKED AT where they are used (part()): building the object is a length per clip and nothing else - a
        # rank of eight reads an eighth of a two-hour stream, and what every rank does for every clip is time no rank count
        # divides (round 5: the per-clip dtype / layout checks moved from here to first use)
        # (``lengths``: the clips' len() when the caller has them already - then ``clips`` is taken as it is, a list of the caller's)
        self._raw = clips if lengths is not None and isinstance(clips, list) else list(clips)
        self._ok = np.zeros(len(self._raw), bool)
        self._have = np.zeros(len(self._raw), np.uint8)  # the clip's address is in _addrs
        n = np.fromiter(map(len, self._raw), np.int64, len(self._raw)) if lengths is None else np.array(lengths, np.int64)
        if len(n) != len(self._raw):
            raise ValueError("one length per clip")
        self.lens = n
        self._addrs = np.zeros(len(n), np.int64)  # filled range by range (addresses): paid where it is used
        self._seen_len = np.zeros(len(n), np.int64)
        self.starts = np.concatenate(([0], np.cumsum(n + gap)))[:-1] if len(n) else np.zeros(0, np.int64)
        self.size = int(n.sum() + gap * max(len(n) - 1, 0))

    def part(self, i: int) -> np.ndarray:
        """Clip ``i`` as contiguous int16 PCM (converted once, on first use; anything that is not integer PCM is refused)."""
        c = self._raw[i]
        if not self._ok[i]:
            if not (type(c) is np.ndarray and c.dtype == np.int16 and c.flags.c_contiguous):
                c = np.asarray(c)
                if c.dtype.kind not in "iu":
                    raise TypeError("JoinedPCM holds int16 PCM")
                c = self._raw[i] = np.ascontiguousarray(c, np.int16)
            self._ok[i] = True
        return c

    @property
    def parts(self) -> List[np.ndarray]:
        return [self.part(i) for i in range(len(self._raw))]

    def addresses(self, c0: int, c1: int) -> np.ndarray:
        """Addresses of the first samples of clips ``c0 .. c1 - 1`` (the staging copies read from there)."""
        new = np.flatnonzero(self._have[c0:c1] == 0) + c0
        if len(new):
            # the contiguous int16 clips of the range in ONE call (csrc/hostext.c: the buffer protocol per clip instead of ~1.6 us of
            # interpreter); what it leaves - another integer dtype, a strided view - goes through part() one by one
            _hostext().scan_pcm16(self._raw, new, self._addrs, self._seen_len, self._have)
            self._ok[new[self._have[new] != 0]] = True
            for i in new[self._have[new] == 0].tolist():
                self._addrs[i] = self.part(i).__array_interface__["data"][0]
                self._seen_len[i] = len(self._raw[i])
                self._have[i] = 1
            if (self._seen_len[new] != self.lens[new]).any():
                raise ValueError("a clip of the joined stream changed its length after the stream was built")
        return self._addrs[c0:c1]

    def __len__(self) -> int:
        return self.size

    def runs(self, a: int, n: int):
        """``(array, first element, count, offset inside [a, a + n))`` for every clip that overlaps ``[a, a + n)``."""
        i = max(int(np.searchsorted(self.starts, a, side="right")) - 1, 0)
        while i < len(self._raw) and self.starts[i] < a + n:
            st = int(self.starts[i])
            lo, hi = max(a, st), min(a + n, st + int(self.lens[i]))
            if hi > lo:
                yield self.part(i), lo - st, hi - lo, lo - a
            i += 1

    def to_array(self) -> np.ndarray:
        out = np.zeros(self.size, np.int16)
        for i, st in enumerate(self.starts.tolist()):
            c = self.part(i)
            out[st:st + len(c)] = c
        return out


def _hostext():
    """The CPython extension next to ``libwwhip.so`` (``csrc/hostext.c``, built by ``wwhip/_build.py``); no fallback."""
    try:
        from . import _wwhostext
    except ImportError as e:  # pragma: no cover
        raise RuntimeError("wwhip/_wwhostext.so is missing: build it with `python __graft_entry__.py`") from e
    return _wwhostext


def _piece_runs(plan: StreamPlan, runs, data: dict, addr: Optional[np.ndarray] = None):
    """For the window runs ``(file, i0, i1)``: the sample range ``[s0, s1)`` each is a function of (its piece), and the copy
    runs that fill the pieces laid end to end: ``(dst_off, src address, count)`` arrays, ascending in ``dst_off``; whatever
    they leave uncovered is zero padding.  A piece of file ``k`` reaches back at most 511 samples into file ``k - 1``.
    Vectorised over the pieces (a test split has thousands); a :class:`JoinedPCM` file expands into one run per clip."""
    ra = np.asarray(runs, np.int64).reshape(-1, 3)
    ks, i0, i1 = ra[:, 0], ra[:, 1], ra[:, 2]
    F = plan.F[ks] if plan.carry else np.zeros_like(ks)
    s0 = 160 * (F + plan.hop * i0)
    s1 = 160 * (F + plan.hop * (i1 - 1) + plan.T - 1) + WINDOW
    soffs = np.concatenate(([0], np.cumsum(s1 - s0))).astype(np.int64)
    d_all, p_all, c_all = [], [], []
    if data is not None and any(isinstance(x, JoinedPCM) for x in data.values()):
        (j,) = data.values()  # (_stage_chunk materialises joined streams that share a list with other files)
        a = (int(plan.pos[0]) if plan.carry else 0) + plan.pad
        st, n = j.starts, j.lens
        for p in range(len(ks)):  # one piece per range of the stream; one run per clip that overlaps it
            lo, hi = max(int(s0[p]), a), min(int(s1[p]), a + len(j))
            if hi > lo and len(n):
                c0 = max(int(np.searchsorted(st, lo - a, side="right")) - 1, 0)
                c1 = int(np.searchsorted(st, hi - a, side="left"))
                q0, q1 = np.maximum(st[c0:c1], lo - a), np.minimum(st[c0:c1] + n[c0:c1], hi - a)
                ok = q1 > q0
                d_all.append((int(soffs[p]) + a - int(s0[p]) + q0)[ok])
                p_all.append((j.addresses(c0, c1) + 2 * (q0 - st[c0:c1]))[ok])
                c_all.append((q1 - q0)[ok])
    else:
        if addr is None:  # (the caller keeps the addresses of the files it has looked at: _prep_chunk)
            addr = np.zeros(len(plan.lengths), np.int64)
            for f, x in data.items():
                addr[f] = x.__array_interface__["data"][0]
        for back in ((1, 0) if plan.carry else (0,)):
            f = ks - back
            ok = f >= 0
            fo = np.where(ok, f, 0)
            a = (plan.pos[fo] if plan.carry else 0) + plan.pad  # stream sample of the file's first sample
            lo, hi = np.maximum(s0, a), np.minimum(s1, a + plan.lengths[fo])
            ok &= hi > lo
            d_all.append((soffs[:-1] + lo - s0)[ok])
            p_all.append((addr[fo] + 2 * (lo - a))[ok])
            c_all.append((hi - lo)[ok])
    d, pp, c = (np.concatenate(x) if x else np.zeros(0, np.int64) for x in (d_all, p_all, c_all))
    order = np.argsort(d, kind="stable")
    return soffs, np.ascontiguousarray(d[order]), np.ascontiguousarray(pp[order]), np.ascontiguousarray(c[order])


_EVAL_LANES = int(os.environ.get("WWHIP_EVAL_LANES", "2"))  # (the variable: development) contexts (HIP streams) the chunks of a pass are dealt to in turn: a chunk's front end and model kernels run beside the next chunk's
_TLS = None  # per host thread (threading.local): {device: the library's uploader - page-locked slots, copy threads, copy stream}
_UPLOAD_SLOTS = 3     # chunks in flight between the interpreter and the kernels: one being written, one uploading, one waiting
# samples of a job's first chunk; sizes double from there (the variable: development).  Round 6, one box, one GPU / one rank's share of
# eight at hey-snips size: 0.5 M 11.2 / 3.09 ms, 1 M 11.0 / 3.17, 2 M 11.0 / 2.91, 4 M 10.6 / 2.59-2.62, 6 M 10.8 / 2.75, 8 M 10.7 / 2.70,
# 24 M 11.4 / 2.74 - a chunk costs the host ~0.1 ms whatever its size, and since a chunk goes up in slices a larger first one no longer
# holds the first kernel back by its whole staging time
_FIRST_CHUNK = int(os.environ.get("WWHIP_FIRST_CHUNK", str(1 << 22)))
_CHUNK_SAMPLES = 24 << 20  # samples staged, uploaded and evaluated per step of the pipeline (48 MB of PCM, ~26 min of audio)


def _uploader(eng: Engine):
    """The calling thread's uploader on the engine's device (one host thread drives an uploader, include/wwhip.h); it goes
    when the thread does."""
    import threading
    from . import _lib
    global _TLS
    if _TLS is None:
        _TLS = threading.local()
    per_thread = _TLS.__dict__.setdefault("uploaders", {})
    up = per_thread.get(eng.ctx.device)
    if up is None or up._h is None:
        up = per_thread[eng.ctx.device] = _lib.Uploader(eng.ctx, _UPLOAD_SLOTS, _COPY_THREADS)
    return up


class _Chunk:
    """One step of a rank's share on its way to the GPU: ``runs`` (file, i0, i1) -> samples in a page-locked slot -> device."""
    __slots__ = ("job", "runs", "n_win", "copy", "keep", "soffs", "foffs", "nf_max", "total_f", "d_pcm", "d_so", "d_fo", "d_wo", "ticket",
                 "host_pieces", "d_mel", "d_out", "lane")


def _prep_chunk(ch: "_Chunk", ph: _Phases) -> None:
    """Host arithmetic of one chunk: which samples of which clips its pieces hold (``ch.copy``: the runs
    ``ww_host_stage_i16`` takes) and the sample / frame offset tables of the pieces.  When a file is not int16 PCM (stereo
    wavs, custom loaders) everything is brought to float32 in [-1, 1) instead (int16 / 32768) and staged through NumPy
    (``ch.host_pieces``)."""
    plan, runs, load = ch.job.plan, ch.runs, ch.job.load
    with ph("slicing"):
        runs = np.asarray(runs, np.int64).reshape(-1, 3)
        ks = np.unique(runs[:, 0])
        files = np.unique(np.concatenate((ks, ks[ks > 0] - 1))) if plan.carry else ks
        raw = getattr(ch.job, "raw_list", None)
        if raw is not None:
            # clips in a Python list (the labelled test set in memory): the files of the chunk that have not been looked at yet in
            # ONE call of the extension (csrc/hostext.c: address, length, "contiguous int16" per clip by the buffer protocol) - no
            # per-clip interpreter work on the way to the copy runs.  Anything else in the list (another dtype, a strided view, a
            # JoinedPCM) switches the job to the per-file path below for good.
            addr, seen_len, have = ch.job._fstate
            new = files[have[files] == 0]
            if len(new):
                if _hostext().scan_pcm16(raw, new, addr, seen_len, have) != len(new):
                    raw = ch.job.raw_list = None
                elif (seen_len[new] != plan.lengths[new]).any():
                    f = int(new[np.flatnonzero(seen_len[new] != plan.lengths[new])[0]])
                    raise ValueError(f"file {f} holds {int(seen_len[f])} samples, the plan was built on {int(plan.lengths[f])}")
        if raw is not None:
            soffs, d, pp, c = _piece_runs(plan, runs, None, addr)
            lens = np.diff(soffs)
            nf = np.where(lens >= WINDOW, (lens - WINDOW) // 160 + 1, 0).astype(np.int64)
            ch.keep = raw  # (the clips must stay where they are until the copy threads have read them: the list holds them)
            ch.copy = (d, pp, c)
            ch.soffs, ch.foffs = soffs, np.concatenate(([0], np.cumsum(nf)))
            ch.nf_max, ch.total_f = int(nf.max()) if len(nf) else 0, int(ch.foffs[-1])
            n_win = np.asarray(ch.n_win, np.int64)
            assert ((n_win == 0) | ((n_win - 1) * plan.hop + plan.T <= nf)).all(), "piece too short for its windows"
            return
        # a file is looked at ONCE per job (it turns up in two chunks at most, and with the carry as its successor's predecessor
        # too): array, address and "is int16 PCM" are kept with the job
        fc = getattr(ch.job, "_files", None)
        if fc is None:
            n_all = len(plan.lengths)
            fc = (([None] * n_all), np.zeros(n_all, np.int64), np.ones(n_all, bool), plan.lengths.tolist())
            if getattr(ch.job, "cache_files", True):
                ch.job._files = fc  # (clips that are in memory anyway; files read from disk are held per chunk only: ch.keep)
        arrs, addr, is_i16, lens_l = fc
        for f in files[addr[files] == 0].tolist():
            if arrs[f] is not None:
                continue
            x = load(f)
            if isinstance(x, JoinedPCM):
                if len(lens_l) > 1:
                    x = x.to_array()  # a joined stream among other files: as an ordinary array
            elif not (type(x) is np.ndarray and x.flags.c_contiguous):
                x = np.ascontiguousarray(x)
            # the copy runs below take their counts from the plan and hand raw addresses to the library's copy threads: a file
            # that holds fewer samples than the plan was built on (a truncated wav whose header says more, a caller's own
            # `lengths`) would be read past its end - and its window counts would be wrong anyway
            if len(x) != lens_l[f]:
                raise ValueError(f"file {f} holds {len(x)} samples, the plan was built on {lens_l[f]}")
            arrs[f] = x
            if isinstance(x, JoinedPCM):
                continue
            if x.dtype == np.int16:
                addr[f] = x.__array_interface__["data"][0]
            else:
                is_i16[f] = False
        data = {f: arrs[f] for f in files.tolist()}
        if not is_i16[files].all():
            # not PCM16 everywhere: float32 samples (librosa's scale) for every file
            data = {f: (x.to_array() if isinstance(x, JoinedPCM) else x) for f, x in data.items()}
            data = {f: (x.astype(np.float32) / np.float32(32768.0) if x.dtype == np.int16 else x.astype(np.float32, copy=False))
                    for f, x in data.items()}
            pieces = []
            for k, i0, i1 in runs.tolist():
                s0, s1 = plan.sample_range(k, i0, i1)
                out = np.zeros(s1 - s0, np.float32)
                for f in ((k - 1, k) if plan.carry and k > 0 else (k,)):
                    a = (int(plan.pos[f]) if plan.carry else 0) + plan.pad
                    lo, hi = max(s0, a), min(s1, a + len(data[f]))
                    if hi > lo:
                        out[lo - s0:hi - s0] = data[f][lo - a:hi - a]
                pieces.append(out)
            ch.host_pieces = pieces
            return
        soffs, d, pp, c = _piece_runs(plan, runs, data, addr)
        lens = np.diff(soffs)
        nf = np.where(lens >= WINDOW, (lens - WINDOW) // 160 + 1, 0).astype(np.int64)
        ch.keep = data  # (the clips must stay where they are until the copy threads have read them)
        ch.copy = (d, pp, c)
        ch.soffs, ch.foffs = soffs, np.concatenate(([0], np.cumsum(nf)))
        ch.nf_max, ch.total_f = int(nf.max()) if len(nf) else 0, int(ch.foffs[-1])
        n_win = np.asarray(ch.n_win, np.int64)
        assert ((n_win == 0) | ((n_win - 1) * plan.hop + plan.T <= nf)).all(), "piece too short for its windows"


def _submit_chunk(eng: Engine, ch: "_Chunk", ph: _Phases) -> None:
    """Hands the chunk to the library's uploader (``ww_uploader_submit``): its thread lays the pieces end to end in a
    page-locked slot - clip samples copied in with streaming stores, the paddings between them zeroed, each sample written
    exactly once - and sends them and the two offset tables to the device on its copy stream.  Returns at once."""
    import torch
    if ch.host_pieces is not None:
        return
    with ph("submit"):
        d, pp, c = ch.copy
        n = len(ch.runs)
        need = int(ch.soffs[-1]) + 16  # (the kernel's vector loads may run a few samples past the end: zeros there too)
        dev = torch.device("cuda", eng.ctx.device)
        ch.d_pcm = torch.empty(need, dtype=torch.int16, device=dev)
        d_meta = torch.empty(3 * (n + 1), dtype=torch.int64, device=dev)
        ch.d_so, ch.d_fo, ch.d_wo = d_meta[:n + 1], d_meta[n + 1:2 * (n + 1)], d_meta[2 * (n + 1):]
        woffs = np.concatenate(([0], np.cumsum(ch.n_win)))  # first window of every piece: the per-clip max needs it (ww_posterior_pick_dev)
        ch.ticket = _uploader(eng).submit(need, d, pp, c, ch.d_pcm.data_ptr(), np.concatenate((ch.soffs, ch.foffs, woffs)), d_meta.data_ptr())
        ch.copy = None  # (ch.keep: the clips stay where they are until the uploader has read them)


def _chunk_forward(eng: Engine, ch: "_Chunk", precise: bool, ph: _Phases) -> None:
    """One front-end launch over the chunk's pieces (each its own framing grid, frame j = samples [160 j, 160 j + 512)) and
    one model launch over their windows (piece p: ``n_win[p]`` windows at rows ``hop i``); the detect rows stay on the device
    (``ch.d_out``) until the job collects them."""
    import torch  # only to hold the device buffers of the batched launch

    dev = torch.device("cuda", eng.ctx.device)
    hop = ch.job.plan.hop
    n = len(ch.n_win)
    ch.d_out = torch.empty((int(np.sum(ch.n_win)), eng.n_out), dtype=torch.float32, device=dev)
    if ch.host_pieces is None:
        # librosa's floats are int16 / 32768 exactly: the device front end divides (correctly rounded) by the same 32768
        with ph("upload_wait"):
            # returns when the chunk's samples are on their way; the library's stream waits for them on the device
            _uploader(eng).wait(ch.ticket, eng.ctx)
            ch.keep = None
        with ph("device_wall"):
            ch.d_mel = torch.empty((max(ch.total_f, 1), eng.n_mel), dtype=torch.float32, device=dev)
            eng.logmel_dev(ch.d_pcm.data_ptr(), ch.d_so.data_ptr(), ch.d_fo.data_ptr(), n, ch.total_f, ch.nf_max, ch.d_mel.data_ptr(),
                           frontend_params(32768.0, False, 0.0, 160, precise))
            foffs, total_f = ch.foffs, ch.total_f
    else:
        with ph("device_wall"):
            mels = eng.logmel(ch.host_pieces, frontend_params(1.0, False, 0.0, 160, precise))
            nf = np.array([len(m) for m in mels], np.int64)
            foffs = np.concatenate(([0], np.cumsum(nf)))
            total_f = int(foffs[-1])
            assert all(w == 0 or (w - 1) * hop + eng.window <= f for f, w in zip(nf, ch.n_win)), "piece too short for its windows"
            ch.d_mel = torch.empty((max(total_f, 1), eng.n_mel), dtype=torch.float32, device=dev)
            ch.d_mel[:total_f] = torch.from_numpy(np.concatenate(mels)).to(dev)
            torch.cuda.current_stream(dev).synchronize()
    with ph("device_wall"):
        # the library picks the tail kernel by the launch's window count; every form of the CRNN associates its sums the same
        # way (csrc/crnn.hip: gru_step), so a posterior does not depend on how many windows its launch holds - nor on how a
        # share is cut into chunks
        eng.forward_segments_dev(ch.d_mel.data_ptr(), total_f, foffs[:-1], np.asarray(ch.n_win, np.int32), hop, ch.d_out.data_ptr())
        # what the flow keeps of a chunk, picked where the rows are: the posterior element of every window (the negative
        # stream) or its maximum over each piece (a wake-word clip) - a14 on the device, 4 or 0.1 bytes per window to keep
        job, n_rows = ch.job, int(ch.d_out.shape[0])
        per_piece = job.eval_type == "false_negatives"
        if job.d_vals is None:
            job.d_vals = torch.empty(job.n_values(job.mine), dtype=torch.float32, device=dev)
            job.val_off = 0
        n_vals = n if per_piece else n_rows
        out = job.d_vals[job.val_off: job.val_off + n_vals]
        job.val_off += n_vals
        if per_piece and ch.d_wo is None:  # (pieces staged through NumPy: their window offsets go up here)
            ch.d_wo = torch.from_numpy(np.concatenate(([0], np.cumsum(ch.n_win))).astype(np.int64)).to(dev)
            torch.cuda.current_stream(dev).synchronize()
        if n_vals:
            eng.posterior_pick_dev(ch.d_out.data_ptr(), n_rows, out.data_ptr(), ch.d_wo.data_ptr() if per_piece else 0, n if per_piece else 0)
    # (nothing here waits for the GPU: ch.d_pcm / ch.d_mel / ch.d_out stay alive until _run_jobs has synchronised)


SHARE_ONLY = "share-only"  # comm_device: run ONE rank's share of a world without a communicator (what a rank of N costs, measured on one GPU)


class _PosteriorJob:
    """One ``get_posterior`` call on this rank: the plan (the same arithmetic on every rank), this rank's share of it cut into
    chunks of about ``_CHUNK_SAMPLES`` samples, and - after :func:`_run_jobs` - the posteriors of the share."""

    def __init__(self, eng: Engine, eval_type: str, test_files, frame_width: int, sample_rate: int, rank: int, world: int,
                 loader, lengths, carry_over: bool, ph: _Phases, info: Optional[dict], first_chunk: Optional[int] = None) -> None:
        self.eng, self.eval_type, self.rank, self.world = eng, eval_type, rank, world
        frame_length = sample_rate // 1000 * frame_width
        in_memory = len(test_files) > 0 and not isinstance(test_files[0], (str, bytes)) and not hasattr(test_files[0], "__fspath__")
        self.cache_files = in_memory  # (_prep_chunk: what it learns about a clip is kept with the job only when the clip stays in memory anyway)
        with ph("plan"):
            self.raw_list = None
            if in_memory:
                self.load = lambda k: test_files[k]  # noqa: E731
                lengths = [len(x) for x in test_files] if lengths is None else lengths
                raw = test_files.raw if isinstance(test_files, _Int16Clips) else test_files
                if isinstance(raw, (list, tuple)) and len(raw) > 1:
                    # (_prep_chunk: per-clip bookkeeping through the extension; the state it keeps per file)
                    self.raw_list = raw
                    self._fstate = (np.zeros(len(raw), np.int64), np.zeros(len(raw), np.int64), np.zeros(len(raw), np.uint8))
            else:
                rd = loader or (lambda p: read_wav_pcm(p, sample_rate))
                self.load = lambda k: rd(str(test_files[k]))  # noqa: E731
                if lengths is None:
                    lengths = [wav_length(str(f), sample_rate) for f in test_files] if loader is None else [len(rd(str(f))) for f in test_files]
            self.plan = plan = StreamPlan(lengths, eng.window, frame_length, sample_rate, 2, carry_over)
            if info is not None:
                info["windows"] = info.get("windows", 0) + plan.total  # inferences of the whole call, all ranks
            self._shares: Optional[List[np.ndarray]] = None
            (self.mine,) = plan.shares_arr(eval_type, world, (rank,))  # (the other ranks' shares: when the gather needs them)
            self.chunks: List[_Chunk] = []
            for runs in self._cut(self.mine, _FIRST_CHUNK if first_chunk is None else first_chunk):
                ch = _Chunk()
                ch.job, ch.runs, ch.n_win = self, runs, runs[:, 2] - runs[:, 1]
                ch.host_pieces = ch.d_pcm = ch.d_mel = ch.d_out = ch.d_wo = ch.ticket = ch.copy = ch.keep = None
                self.chunks.append(ch)
        self.vals: Optional[np.ndarray] = None  # the share's values on the host (tests feed them in; finish() falls back to them)
        self.d_vals = None                       # ... on the device: one per window (negative stream) / per run (wake-word clips)
        self.val_off = 0

    def _cut(self, runs: np.ndarray, first: int) -> List[np.ndarray]:
        """The share (``[n, 3]`` runs) as consecutive groups of runs; a run longer than a chunk (the negative stream is ONE
        file) is cut into window ranges - each re-reads the ``T - hop`` frames it shares with its neighbour.  Chunk sizes
        (samples) grow from ``first`` by doubling up to ``_CHUNK_SAMPLES``: a small first chunk is on the GPU early, large
        later ones keep the launches efficient.  A share of many short runs (the wake-word clips: thousands) is grouped by
        its cumulative sample counts, one step per CHUNK; only runs that need cutting are walked one by one."""
        plan = self.plan
        per_win = 160 * plan.hop
        size = max(min(first, _CHUNK_SAMPLES), per_win)
        if len(runs) == 0:
            return []
        nw = runs[:, 2] - runs[:, 1]
        # (the wake-word clips' share keeps one value per run - the clip's maximum, picked on the device - so its runs are never cut:
        # a clip longer than a chunk is a chunk of its own)
        if int(nw.max()) <= (size + size // 2) // per_win or self.eval_type == "false_negatives":  # no run is ever cut: sizes only grow
            c = np.cumsum(160 * (plan.hop * (nw - 1) + plan.T - 1) + WINDOW)
            out, j0, base = [], 0, 0
            while j0 < len(runs):
                j1 = max(int(np.searchsorted(c, base + size, side="right")), j0 + 1)  # (a chunk holds at least one run)
                out.append(runs[j0:j1])
                base, j0 = int(c[j1 - 1]), j1
                size = min(2 * size, _CHUNK_SAMPLES)
            return out
        out, cur, cur_n = [], [], 0

        def close():
            nonlocal cur, cur_n, size
            if cur:
                out.append(np.array(cur, np.int64).reshape(-1, 3))
                cur, cur_n = [], 0
                size = min(2 * size, _CHUNK_SAMPLES)

        for k, i0, i1 in runs.tolist():
            while i1 - i0 > (size + size // 2) // per_win:  # (a remainder below half a chunk stays with the last cut)
                close()
                w = max(size // per_win, 1)
                cur, cur_n = [(k, i0, i0 + w)], size
                i0 += w
                close()
            n = 160 * (plan.hop * (i1 - i0 - 1) + plan.T - 1) + WINDOW
            if cur and cur_n + n > size:
                close()
            cur.append((k, i0, i1))
            cur_n += n
        close()
        return out

    @property
    def shares(self) -> List[np.ndarray]:
        """Every rank's share (the gather places any rank's values by them), worked out on first use - :func:`_run_jobs` asks for it
        once everything is launched, i.e. while the GPU is still busy with this rank's chunks."""
        if self._shares is None:
            others = [r for r in range(self.world) if r != self.rank]
            rest = self.plan.shares_arr(self.eval_type, self.world, others)
            self._shares = [None] * self.world  # type: ignore[list-item]
            self._shares[self.rank] = self.mine
            for r, sh in zip(others, rest):
                self._shares[r] = sh
        return self._shares

    def slots_of(self, runs: np.ndarray):
        """Global slot of every window of the runs: offs[k] + i0 .. offs[k] + i1 - 1, run after run."""
        if len(runs) == 0:
            return np.zeros(0, np.int64)
        nw = runs[:, 2] - runs[:, 1]
        first = self.plan.offs[runs[:, 0]] + runs[:, 1]
        ends = np.cumsum(nw)
        if (first[1:] == first[:-1] + nw[:-1]).all():
            return slice(int(first[0]), int(first[0] + ends[-1]))  # consecutive slots (one rank; a contiguous range of the stream)
        return np.arange(int(ends[-1]), dtype=np.int64) + np.repeat(first - (ends - nw), nw)

    def share_samples(self) -> int:
        """Samples this rank's share is a function of (what it stages and uploads)."""
        nw = self.mine[:, 2] - self.mine[:, 1]
        return int((160 * (self.plan.hop * (nw - 1) + self.plan.T - 1) + WINDOW).sum()) if len(nw) else 0

    def n_values(self, share: np.ndarray) -> int:
        """Values a rank contributes: one per run of its share for the wake-word clips (the maximum over the run's windows,
        ``evaluate_models.py:98-99``), one per window for the negative stream."""
        return len(share) if self.eval_type == "false_negatives" else int((share[:, 2] - share[:, 1]).sum())

    def collect(self, ph: _Phases) -> None:
        """After :func:`_run_jobs` (every kernel has finished): the share's values stay on the device (``d_vals``); the
        detect rows go."""
        import torch
        if self.d_vals is None:  # (a share without a window)
            self.d_vals = torch.empty(0, dtype=torch.float32, device=torch.device("cuda", self.eng.ctx.device))
        for ch in self.chunks:
            ch.d_out = None

    def finish_dev(self, comm_device: Optional[str], ph: _Phases):
        """The gather (the one exchange).  Returns, on every rank, what ``get_posterior`` returns - the wake-word clips' maxima
        as a host array (one float per file), the negative stream's posteriors as a tensor on the device that produced them
        (smoothing and sweep read them there: :func:`evaluate_reference_flow_sharded`)."""
        import torch
        plan = self.plan
        per_run = self.eval_type == "false_negatives"
        every = self.gather_vals(comm_device, ph)
        vals = every[self.rank]
        with ph("gather"):
            if per_run:
                return self.place_host(every, comm_device)
            if self.world == 1:
                return vals  # (one rank: its windows are the stream, in order)
            post_t = torch.zeros(plan.total, dtype=torch.float32, device=vals.device)
            for sh, v in zip(self.shares, every):
                if v is not None and len(sh):
                    post_t[self.slots_of(sh)] = v  # (a rank's windows are a contiguous range of the stream: a slice)
            return post_t

    def gather_vals(self, comm_device: Optional[str], ph: _Phases) -> list:
        """The one exchange: every rank's values as tensors where this rank's are (``None`` for a rank that was not run: SHARE_ONLY)."""
        import torch
        vals = self.d_vals if self.d_vals is not None else torch.from_numpy(np.ascontiguousarray(self.vals, np.float32))
        with ph("gather"):
            if self.eval_type == "false_negatives" and (self.plan.n_win == 0).any():
                raise ValueError("max() arg is an empty sequence")  # an empty clip: what np.max raises in the reference's loop
            every: list = [None] * self.world
            if self.world > 1 and comm_device != SHARE_ONLY:
                # the plan is the same arithmetic on every rank, so every rank knows which slots any rank's values fill: the
                # exchange is the values alone (float32, padded to the largest share)
                from . import dist as D
                every = D.gather_values_t(vals, [self.n_values(sh) for sh in self.shares], device=comm_device)
            else:
                every[self.rank] = vals  # (SHARE_ONLY: one rank's share timed without its peers - their slots stay zero)
        return every

    def place_host(self, every: list, comm_device: Optional[str]) -> np.ndarray:
        """The wake-word clips' maxima per FILE on the host from the gathered values (one float per clip; 2,529 at hey-snips size)."""
        post = np.full(len(self.plan.lengths), -np.inf if self.world == 1 or comm_device != SHARE_ONLY else 0.0, np.float32)
        for r, v in enumerate(every):
            if v is not None:
                sh = self.mine if r == self.rank else self.shares[r]
                if len(sh):
                    np.maximum.at(post, sh[:, 0], v.cpu().numpy())
        return post

    def finish(self, comm_device: Optional[str], ph: _Phases, as_array: bool):
        """:meth:`finish_dev` on the host: what ``get_posterior`` returns, on every rank."""
        post = self.finish_dev(comm_device, ph)
        if not isinstance(post, np.ndarray):
            with ph("d2h"):
                post = post.cpu().numpy()
        return post if as_array else post.tolist()


def _run_jobs(eng: Engine, jobs: Sequence, precise: bool, ph: _Phases, timing: Optional[dict]) -> List[_PosteriorJob]:
    """The chunks of the jobs, in order, through a pipeline of three workers that never wait for one another's locks: THIS
    thread works out which samples a chunk's pieces hold (:func:`_prep_chunk`, Python arithmetic) and hands them to the
    library's uploader (:func:`_submit_chunk`); the uploader's threads write the chunk into page-locked memory and start its
    upload; and as soon as a chunk is on its way this thread enqueues the front end and the model over it behind the upload
    (:func:`_chunk_forward`: a device-side wait, no call waits for the GPU) - the interpreter, the copy threads, the DMA
    engine and the kernels work on consecutive chunks at the same time.  ``jobs``: :class:`_PosteriorJob` objects or callables
    that build one (or ``None``); a callable runs when the pipeline gets there, so a second job is joined and planned while
    the first one's chunks upload and compute.  Returns the jobs, their shares' posteriors collected."""
    from collections import deque

    made: List[_PosteriorJob] = []
    pending: "deque[_Chunk]" = deque()   # submitted, not yet launched
    done: List[_Chunk] = []
    # Consecutive chunks go to alternating lanes (Engine.lane: the same model on contexts of their own): a chunk is a front-end
    # launch, a model launch of two or three kernels and a pick, each a chip-wide grid of equal workgroups that start and end
    # together - on ONE stream the GPU drains between them.  Two chunks in flight fill each other's ramps and tails (the clip
    # path's pipelined contexts: 57 -> 44 us per step).  A chunk's values land in its own slice of the job's buffer, so the
    # lanes never touch the same bytes; every lane is synchronised before anything is read.
    lanes = [eng.lane(k) for k in range(max(1, _EVAL_LANES))] if hasattr(eng, "lane") else [eng]
    n_launched = 0

    def forward(ch0: "_Chunk") -> None:
        nonlocal n_launched
        ch0.lane = lanes[n_launched % len(lanes)]
        n_launched += 1
        done.append(ch0)
        _chunk_forward(ch0.lane, ch0, precise, ph)

    def sync_all() -> None:
        for le in lanes:
            le.ctx.synchronize()

    # timing["kernel_times"] = False: the host phases only (perf_counter reads) - no HIP events around the launches, no read-back
    # of them: what a pass costs when nobody looks at its kernels (bench.py times such passes and profiles one more)
    kernel_times = timing is not None and timing.get("kernel_times", True)
    if kernel_times:
        for le in lanes:
            le.ctx.profile(True)
    try:
        for j in jobs:
            job = j() if callable(j) else j
            if job is None:
                continue
            made.append(job)
            for ch in job.chunks:
                _prep_chunk(ch, ph)
                _submit_chunk(eng, ch, ph)
                pending.append(ch)
                # launch what is ready; never run more than the uploader's slots ahead of the launches
                while pending and (len(pending) >= _UPLOAD_SLOTS or pending[0].ticket is None or _uploader(eng).done(pending[0].ticket)):
                    forward(pending.popleft())
        while pending:
            forward(pending.popleft())
        with ph("plan"):
            for job in made:  # every rank's share, for the gather: worked out now, beside the kernels that are still running
                job.shares
        with ph("device_wall"):
            sync_all()
    finally:
        # on the way out of an error too: nothing enqueued or submitted may outlive its buffers or the clips it reads
        for ch in pending:
            if ch.ticket is not None:
                try:
                    _uploader(eng).wait(ch.ticket, eng.ctx)
                except Exception:  # noqa: BLE001 - the first error is the one that is reported
                    pass
        if done or pending:
            sync_all()
        for ch in list(done) + list(pending):
            ch.d_pcm = ch.d_mel = ch.d_so = ch.d_fo = ch.keep = None
    if kernel_times:
        with ph("profile_read"):
            profs = []
            for le in lanes:
                profs.append(le.ctx.profile_read())
                le.ctx.profile(False)
        for prof in profs:
            timing["device_ms"] = timing.get("device_ms", 0.0) + sum(v["total_ms"] for v in prof.values())
            for k, v in prof.items():
                timing.setdefault("kernels_ms", {})[k] = timing.get("kernels_ms", {}).get(k, 0.0) + v["total_ms"]
    if timing is not None:
        timing["chunks"] = timing.get("chunks", 0) + len(done)
    for job in made:
        job.collect(ph)
    return made


def get_posterior_sharded(models_dir, model_type, eval_type, test_files, frame_width, sample_rate, rank: int = 0,
                          world: int = 1, comm_device: Optional[str] = None,
                          loader: Optional[Callable[[str], np.ndarray]] = None, lengths: Optional[Sequence[int]] = None,
                          carry_over: bool = True, device: int = 0, engine: Optional[Engine] = None,
                          precise: bool = True, timing: Optional[dict] = None, as_array: bool = False,
                          info: Optional[dict] = None):
    """:func:`get_posterior` with its windows dealt to ``world`` ranks (one process per GPU; ``torch.distributed``
    initialised by the caller when ``world > 1``).  ``"false_negatives"``: whole files, longest first round-robin;
    ``"false_accepts"``: the window list - for the reference's evaluator ONE long wav (``evaluate_models.py:317-321``) -
    cut into ``world`` contiguous posterior ranges (``dist.split_stream``).  A rank loads and front-ends only the samples
    its windows are functions of: posterior ``i`` of a file needs the global frames ``[F + 2 i, F + 2 i + T)``, i.e. each
    range re-reads a ``T - 2``-frame overlap and the results are exact.  The share goes to the GPU in chunks (4 M samples
    doubling up to ``_CHUNK_SAMPLES`` = 24 M, ~26 minutes of audio) through the library's uploader (:func:`_run_jobs`).  The one
    exchange is the posterior gather.  Every rank returns the full list :func:`get_posterior` returns.

    ``test_files``: paths (default loader :func:`read_wav_pcm`; ``lengths`` default = wav headers) or arrays already in
    memory (int16 PCM, or float32 samples in [-1, 1)).  ``precise=False``: the fp32-FFT front end (``ww_frontend_params.precise``
    = 0) instead of the reference's float64 STFT.  ``timing``: a dict that receives this rank's wall-clock seconds per host
    phase - ``plan`` (lengths, shares, chunk cuts), ``slicing`` (which samples of which clips a chunk holds), ``submit`` (handing a
    chunk to ``ww_uploader``: its threads stage and upload it beside everything else), ``upload_wait`` (this thread waiting for a
    submitted chunk's copies to be enqueued), ``device_wall`` (launches + the final wait for the GPU), ``gather``, ``d2h``,
    ``profile_read`` - plus ``device_ms`` / ``kernels_ms`` (HIP events around every kernel of the call, ``ww_profile_read``) and
    ``chunks``; ``info`` receives ``windows`` (inferences of the whole call, all ranks).  ``as_array``: a float32 array
    instead of the reference's list (two hours of negatives are 360,000 Python floats)."""
    if model_type not in ("CRNN", "Wavenet"):
        raise ValueError("model_type must be 'CRNN' or 'Wavenet'")
    if eval_type not in ("false_negatives", "false_accepts"):
        raise ValueError("eval_type must be 'false_negatives' or 'false_accepts'")
    if len(test_files) == 0:
        return []
    eng: Engine = engine or engine_for(models_dir, device)
    ph = _Phases(timing)
    (job,) = _run_jobs(eng, [lambda: _PosteriorJob(eng, eval_type, test_files, frame_width, sample_rate, rank, world, loader,
                                                   lengths, carry_over, ph, info)], precise, ph, timing)
    return job.finish(comm_device, ph, as_array)


def join_negatives(clips: Sequence[np.ndarray], num_files: int, sample_rate: int = 16000) -> np.ndarray:
    """``concatenate_FA`` (``evaluate_models.py:150-160``) on PCM in memory: ``clips[0] + (100 ms silence + clip) for
    clips[1:num_files]``."""
    gap = np.zeros(sample_rate // 10, np.int16)
    parts = [np.asarray(clips[0], np.int16)]
    for c in clips[1:max(num_files, 0)]:
        parts += [gap, np.asarray(c, np.int16)]
    return np.concatenate(parts)


def join_negatives_lazy(clips: Sequence[np.ndarray], num_files: int, sample_rate: int = 16000) -> JoinedPCM:
    """:func:`join_negatives` as a :class:`JoinedPCM` (no copy of the clips)."""
    return JoinedPCM(clips[:max(num_files, 1)], sample_rate // 10)


def evaluate_negative_stream_sharded(engine: Engine, stream_pcm, rank: int = 0, world: int = 1,
                                     comm_device: Optional[str] = None, precise: bool = True,
                                     timing: Optional[dict] = None, info: Optional[dict] = None) -> np.ndarray:
    """The reference's false-accept leg (``evaluate_models.py:317-321``: ``get_posterior(..., "false_accepts",
    [FAR_path])``) on one long PCM stream (an int16 array or a :class:`JoinedPCM`), cut into ``world`` contiguous posterior
    ranges; full posterior array on every rank (smoothing across the cuts happens after the gather, :func:`far_frr`)."""
    stream = stream_pcm if isinstance(stream_pcm, JoinedPCM) else np.asarray(stream_pcm)
    return get_posterior_sharded(engine.model_dir, "CRNN" if engine.is_crnn else "Wavenet", "false_accepts", [stream], 20, 16000,
                                 rank, world, comm_device, engine=engine, precise=precise, timing=timing, as_array=True, info=info)


class _Int16Clips:
    """Labelled clips in memory as the int16 PCM the reference flow feeds (``evaluate_models.py:45-61``), each converted when it
    is first asked for - by the rank that stages it."""

    def __init__(self, raw: list) -> None:
        self.raw = raw

    def __len__(self) -> int:
        return len(self.raw)

    def __getitem__(self, k: int) -> np.ndarray:
        c = self.raw[k]
        return c if type(c) is np.ndarray and c.dtype == np.int16 else np.asarray(c, np.int16)


def evaluate_reference_flow_sharded(engine: Engine, clips: Sequence[np.ndarray], labels: Sequence[int], rank: int = 0,
                                    world: int = 1, comm_device: Optional[str] = None, thresholds=None, windowsize: int = 30,
                                    precise: bool = True, timing: Optional[dict] = None):
    """``utils/evaluate_models.py`` ``main()`` (``:281-326``) on labelled int16 clips in memory, sharded over ``world``
    ranks: the wake-word clips go file by file through one never-reset ``Filter`` (quirk C2; utterance-sharded, a rank
    re-reads the <= 511-sample tail of a file's predecessor), the first ``num_wakewords`` other clips are joined by 100 ms
    of silence into ONE stream (``concatenate_FA``) that is evaluated continuously and cut into contiguous posterior ranges
    (:func:`evaluate_negative_stream_sharded`), hours = the joined stream's duration; rank 0 smooths and sweeps.
    Returns the result dict on rank 0 and ``None`` elsewhere.  ``timing``: see :func:`get_posterior_sharded`; the two legs
    add up in it, and ``sweep`` is rank 0's smoothing + threshold sweep."""
    ph = _Phases(timing)
    labels = np.asarray(labels).astype(bool)
    num_wakewords = int(labels.sum())
    info: dict = {}
    made: dict = {}

    # both legs as ONE pipeline (:func:`_run_jobs`), the negative stream FIRST: it holds more windows per uploaded byte (no
    # half seconds of padding around every clip), so the GPU has work for the time the wake-word clips take to stage
    def negative_job():
        with ph("prepare"):
            # the clips the stream joins and their lengths in ONE call of the extension (csrc/hostext.c: take) - what every rank needs
            # of every clip, whatever the world size
            other, lens = _hostext().take(clips, np.flatnonzero(~labels)[:max(num_wakewords, 1)].astype(np.int64))
            made["n_joined"] = len(other)
            stream = made["stream"] = JoinedPCM(other, 16000 // 10, np.frombuffer(lens, np.int64)) if other else None
        if stream is None or len(stream) == 0:
            return None
        made["neg"] = _PosteriorJob(engine, "false_accepts", [stream], 20, 16000, rank, world, None, None, True, ph, info)
        return made["neg"]

    def wake_job():
        if num_wakewords == 0:
            return None
        with ph("prepare"):
            # what every rank needs of every clip is its LENGTH (the plan); a clip itself is looked at by the rank whose share
            # holds it, when its chunk is staged (_prep_chunk)
            raw, lens = _hostext().take(clips, np.flatnonzero(labels).astype(np.int64))
            wake = _Int16Clips(raw)
            lengths = np.frombuffer(lens, np.int64)
        # full-size chunks at once when the negative share keeps the pipeline busy for a while (two full chunks or more: one GPU at
        # hey-snips size); a rank of eight holds a ninth of that - its wake-word clips start small again, so that the first of
        # them are on the GPU while the rest are still being staged (one 27 MB chunk: stage, upload, compute, one after the other)
        neg = made.get("neg")
        busy = neg is not None and neg.share_samples() >= 2 * _CHUNK_SAMPLES
        made["wake"] = _PosteriorJob(engine, "false_negatives", wake, 20, 16000, rank, world, None, lengths, True, ph, info,
                                     first_chunk=_CHUNK_SAMPLES if busy else None)
        return made["wake"]

    _run_jobs(engine, [negative_job, wake_job], precise, ph, timing)
    import torch
    wake = made.get("wake")
    every = wake.gather_vals(comm_device, ph) if wake is not None else []
    neg_t = made["neg"].finish_dev(comm_device, ph) if "neg" in made else None
    stream = made.get("stream")
    if rank != 0:
        return None
    hours = (len(stream) if stream is not None else 0) / 16000.0 / 3600.0
    thr = default_thresholds() if thresholds is None else np.asarray(thresholds, np.float64)
    pos = None
    if neg_t is not None and neg_t.is_cuda and len(neg_t):
        # smoothing + sweep where the gathered values are (ww_far_frr_dev): the thresholds go up, 2 x 100 counters come back.  The
        # positives are counted where they are as well: one value per wake-word clip (its runs are never cut, _PosteriorJob._cut),
        # as the pick kernels - or the gather - left them; their order does not matter to a count.  The per-file array the caller
        # gets is put together afterwards.
        there = [v for v in every if v is not None and v.numel()]
        with ph("sweep"):
            if there and all(v.is_cuda and v.device == neg_t.device for v in there):
                d_pos = there[0] if len(there) == 1 else torch.cat(there)
            else:
                pos = wake.place_host(every, comm_device) if wake is not None else np.zeros(0, np.float32)
                d_pos = torch.from_numpy(pos).to(neg_t.device)
            if wake is None or wake.world > 1 or pos is not None:
                torch.cuda.current_stream(neg_t.device).synchronize()  # (the gather's copies, a cat, an upload: done before the library's stream reads)
            frr, fa, cnt = engine.far_frr_dev(d_pos.data_ptr(), int(d_pos.numel()), neg_t.data_ptr(), len(neg_t), thr, float(max(num_wakewords, 1)),
                                              hours, windowsize)
        with ph("d2h"):
            if pos is None:
                pos = wake.place_host(every, comm_device)  # (for the caller: the curves above did not need them on the host)
            neg = neg_t.cpu().numpy()
    else:
        pos = wake.place_host(every, comm_device) if wake is not None else np.zeros(0, np.float32)
        neg = np.zeros(0, np.float32) if neg_t is None else neg_t.cpu().numpy()
        with ph("sweep"):
            thr, frr, fa, cnt = far_frr(pos, neg, max(num_wakewords, 1), hours, thr, windowsize, engine=engine)
    return {"thresholds": thr, "frr": frr, "fa_per_hour": fa, "fa_count": cnt, "frr_at_0.5_fa_per_hour": frr_at_fa(frr, fa, 0.5),
            "positives": pos, "negatives": neg, "hours": hours, "num_wakewords": num_wakewords,
            "negative_clips_joined": made.get("n_joined", 0), "windows": int(info.get("windows", 0)),
            "negative_windows": int(len(neg)),
            "posterior_checksum": float(neg.sum(dtype=np.float64) + pos.sum(dtype=np.float64))}


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


def open_h5(data_file: str):
    """h5py when it is installed, otherwise the built-in reader (:mod:`wwhip.h5min`) - both give
    ``keys()``, ``[name][()]`` and ``.attrs`` for the reference's feature files."""
    try:
        import h5py  # type: ignore
        return h5py.File(data_file, "r")
    except ImportError:
        from . import h5min
        return h5min.File(data_file)


def load_h5(data_file: str, timesteps: int, num_features: int):
    """The reference's on-disk format (one dataset per clip, attr ``is_hotword``,
    ``filter_dataset_to_h5.py:136-145``) -> ``(X [N, timesteps, F], labels)`` exactly as
    ``evaluate_tf_lite_opts.py:35-47`` builds them (keys in h5py's name order)."""
    feats, labels = [], []
    with open_h5(data_file) as h5:
        for key in h5.keys():
            labels.append(int(h5[key].attrs["is_hotword"]))
            feats.append(h5[key][()])
    return load_data(feats, labels, timesteps, num_features)


def models_predict(engine: Engine, X: np.ndarray, threshold: float = 0.5) -> Tuple[List[int], np.ndarray]:
    """One window per clip, class 1 when posterior >= threshold (``evaluate_tf_lite_opts.py:49-69``)."""
    post = engine.forward(X)[:, engine.posterior_index]
    return [1 if p >= threshold else 0 for p in post], post


# ----------------------------------------------------------------------------------------------
# Whole-test-set evaluation on in-memory clips (SURVEY 8d cfg 1 / cfg 4): the a17 flow (one window
# per clip) and the a14-a16 flow (sliding hop 2, max per positive clip, negatives as one stream,
# smoothing + sweep), batched: ONE front-end launch for all clips, ONE model launch for all
# windows of all clips, one sweep.
# ----------------------------------------------------------------------------------------------
def synth_clip(rng: np.random.Generator, n: int, noise: float = 2000.0, chirp: float = 8000.0) -> np.ndarray:
    """SURVEY 8(d) cfg-1 stand-in for a hey-snips clip: Gaussian noise + linear chirp 200->4000 Hz."""
    t = np.arange(n) / 16000.0
    dur = max(n / 16000.0, 1e-3)
    phase = 2 * np.pi * (200.0 * t + 0.5 * (4000.0 - 200.0) / dur * t * t)
    x = rng.normal(0.0, noise, n) + chirp * np.sin(phase)
    return np.clip(np.rint(x), -32768, 32767).astype(np.int16)


def synth_testset(n_clips: int = 2048, seed: int = 1234, min_s: float = 0.8, max_s: float = 2.5):
    rng = np.random.default_rng(seed)
    lens = rng.integers(int(min_s * 16000), int(max_s * 16000) + 1, n_clips)
    clips = [synth_clip(rng, int(n)) for n in lens]
    labels = (rng.random(n_clips) < 0.1).astype(np.uint8)
    return clips, labels


def synth_testset_scaled(n_wake: int = 2529, n_other: int = 2529, seed: int = 4321, min_s: float = 0.8, max_s: float = 2.5):
    """A stand-in at the SIZE of the hey-snips test split as the reference's evaluator uses it (SURVEY 8d cfg 4): ``n_wake``
    wake-word clips and the first ``n_other`` of the other clips (``utils/evaluate_models.py:299`` joins exactly
    ``num_wakewords`` of them, whatever the split holds), the same signal model as :func:`synth_clip` generated in float32
    (a two-hour set in seconds).  Returns ``(clips, labels)`` with the wake-word clips first."""
    rng = np.random.default_rng(seed)
    n = n_wake + n_other
    lens = rng.integers(int(min_s * 16000), int(max_s * 16000) + 1, n)
    clips = []
    two_pi = np.float32(2 * np.pi)
    for m in lens:
        m = int(m)
        t = np.arange(m, dtype=np.float32) * np.float32(1.0 / 16000.0)
        dur = np.float32(max(m / 16000.0, 1e-3))
        x = rng.standard_normal(m, dtype=np.float32) * np.float32(2000.0)
        x += np.float32(8000.0) * np.sin(two_pi * (np.float32(200.0) * t + np.float32(0.5 * 3800.0) / dur * t * t))
        clips.append(np.clip(np.rint(x), -32768, 32767).astype(np.int16))
    labels = np.zeros(n, np.uint8)
    labels[:n_wake] = 1
    return clips, labels


def clip_posteriors(engine: Engine, clips: Sequence[np.ndarray], hop: int = 2, fp=None):
    """Per clip: posterior of the single end-padded window (a17) and the sliding posteriors
    (hop 2) of the clip padded by 0.5 s of zeros on both sides (evaluate_models.py:52-53), ring
    reset per clip.  Returns (one_window [N], sliding list of arrays).

    One pass on the device: the padded clips go up once, one front-end launch makes their log-mel rows, and one
    model launch evaluates every window.  The 0.5 s of leading zeros are exactly 50 hops, so frame k of the bare
    clip is row k + 50 of the padded one: the single window of a17 is rows [50, 50 + min(frames, T)) of the same
    buffer (``valid`` < T zero-pads it), no second front-end pass and no mel round trip through the host."""
    import torch  # only to hold the device buffers of the batched launch

    fp = fp or frontend_params()
    pidx = engine.posterior_index
    T, hop_s, PAD = engine.window, int(fp.hop), CLIP_PAD
    n = len(clips)
    if n == 0:
        return np.zeros(0, np.float32), []
    if PAD % hop_s:
        raise ValueError(f"front-end hop {hop_s} does not divide the 0.5 s padding")
    lens = np.array([len(c) for c in clips], np.int64)
    soffs = np.concatenate(([0], np.cumsum(lens + 2 * PAD)))
    # staged in page-locked memory (kept for the next call): the upload of the padded batch is the largest single
    # cost of this function, and from pageable memory it runs at a fraction of the bus rate
    global _PIN
    need = int(soffs[-1]) + 16
    if _PIN is None or _PIN.numel() < need:
        _PIN = torch.empty(need + need // 4, dtype=torch.int16, pin_memory=True)
    pin = _PIN[:need]
    pcm = pin.numpy()
    pcm[need - 16:] = 0
    for i, c in enumerate(clips):
        a = int(soffs[i])
        pcm[a: a + PAD] = 0
        pcm[a + PAD: a + PAD + len(c)] = c
        pcm[a + PAD + len(c): a + 2 * PAD + len(c)] = 0
    nf_pad = np.where(lens + 2 * PAD >= 512, (lens + 2 * PAD - 512) // hop_s + 1, 0)
    nf_bare = np.where(lens >= 512, (lens - 512) // hop_s + 1, 0)
    foffs = np.concatenate(([0], np.cumsum(nf_pad)))
    total_f = int(foffs[-1])
    # windows: one per clip (rows of the bare clip), then the sliding ones of the padded clip
    nw = np.where(nf_pad >= T, (nf_pad - T) // hop + 1, 0)
    woffs = np.concatenate(([0], np.cumsum(nw)))
    # the n single windows (ragged: valid < T) as an explicit list; the sliding ones as n sequences of one buffer, which
    # lets the CRNN compute every time position once per clip instead of once per window (ww_forward_segments_dev)
    win_row = (foffs[:-1] + PAD // hop_s).astype(np.int64)
    win_valid = np.minimum(nf_bare, T).astype(np.int32)
    n_slide = int(woffs[-1])
    dev = torch.device("cuda", engine.ctx.device)  # the engine's GPU, whatever torch's current device is
    d_pcm = pin.to(dev, non_blocking=True)
    d_so, d_fo = torch.from_numpy(soffs).to(dev), torch.from_numpy(foffs).to(dev)
    d_mel = torch.empty((max(total_f, 1), engine.n_mel), dtype=torch.float32, device=dev)
    d_row, d_valid = torch.from_numpy(win_row).to(dev), torch.from_numpy(win_valid).to(dev)
    d_out = torch.empty((n + n_slide, engine.n_out), dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    engine.logmel_dev(d_pcm.data_ptr(), d_so.data_ptr(), d_fo.data_ptr(), n, total_f, int(nf_pad.max()), d_mel.data_ptr(), fp)
    # the n single windows: one fused kernel or front + tail kernels by n - the same bits either way (one association of every
    # sum in all forms of the CRNN), so the sharded evaluation does not differ between world sizes
    engine.forward_windows_dev(d_mel.data_ptr(), total_f, d_row.data_ptr(), d_valid.data_ptr(), n, d_out.data_ptr())
    if n_slide:
        engine.forward_segments_dev(d_mel.data_ptr(), total_f, foffs[:-1], nw, hop, d_out[n:].data_ptr())
    engine.ctx.synchronize()
    post = d_out.cpu().numpy()[:, pidx]
    p_one, slide = post[:n], post[n:]
    return p_one, [slide[woffs[i]:woffs[i + 1]] for i in range(n)]


def evaluate_testset(engine: Engine, clips: Sequence[np.ndarray], labels: Sequence[int], thresholds=None,
                     windowsize: int = 30, fp=None):
    """FRR / FA-per-hour curves + FRR@0.5FA/h + one-window accuracy for a labelled clip set."""
    labels = np.asarray(labels).astype(bool)
    p_one, sliding = clip_posteriors(engine, clips, fp=fp)
    pos = np.array([s.max() if len(s) else 0.0 for s, l in zip(sliding, labels) if l], np.float32)
    neg = np.concatenate([s for s, l in zip(sliding, labels) if not l]) if (~labels).any() else np.zeros(0, np.float32)
    hours = sum((len(c) + 16000) for c, l in zip(clips, labels) if not l) / 16000.0 / 3600.0
    thr, frr, fa, cnt = far_frr(pos, neg, max(int(labels.sum()), 1), hours, thresholds, windowsize, engine=engine)
    preds = (p_one >= 0.5)
    return {"thresholds": thr, "frr": frr, "fa_per_hour": fa, "fa_count": cnt, "frr_at_0.5_fa_per_hour": frr_at_fa(frr, fa, 0.5),
            "one_window_posteriors": p_one, "one_window_accuracy": float((preds == labels).mean()),
            "positives": pos, "negatives": neg, "hours": hours}


def evaluate_testset_sharded(engine: Engine, clips: Sequence[np.ndarray], labels: Sequence[int], rank: int = 0,
                             world: int = 1, comm_device: Optional[str] = None, thresholds=None, windowsize: int = 30,
                             fp=None):
    """SURVEY 8(d) cfg 4, per-clip variant: :func:`evaluate_testset` with the utterances dealt longest-first round-robin to
    ``world`` ranks (one process per GPU, ``torch.distributed`` already initialised by the caller when ``world > 1``).  Each
    rank runs its own clips; the one exchange is the posterior gather; rank 0 smooths and sweeps.  A posterior is a function
    of its own window's mel rows only (``utils/evaluate_models.py:70-88``; the CRNN's sliding form computes a time position
    once per CLIP - ``ww_forward_segments_dev`` - never across clips), so the result is identical for every ``world``.
    (The reference's own flow - one continuous negative stream - is :func:`evaluate_reference_flow_sharded`.)
    Returns the result dict on rank 0 and ``None`` elsewhere."""
    from . import dist as D
    labels = np.asarray(labels).astype(bool)
    T, n = engine.window, len(clips)
    hop_s = int(fp.hop) if fp is not None else 160
    # global layout of the sliding posteriors: pure arithmetic, identical on every rank (the same PAD / front-end hop / window
    # hop clip_posteriors uses; asserted per clip below)
    n_frames = np.array([(((len(c) + 2 * CLIP_PAD) - WINDOW) // hop_s + 1 if len(c) + 2 * CLIP_PAD >= WINDOW else 0) for c in clips], np.int64)
    n_win = np.where(n_frames >= T, (n_frames - T) // CLIP_HOP + 1, 0)
    offs = np.concatenate(([0], np.cumsum(n_win)))
    mine = D.shard_by_length([len(c) for c in clips], world)[rank]
    p_one, sliding = clip_posteriors(engine, [clips[i] for i in mine], CLIP_HOP, fp)
    for j, i in enumerate(mine):
        if len(sliding[j]) != n_win[i]:
            raise RuntimeError(f"clip {i}: {len(sliding[j])} sliding posteriors, the global layout expects {n_win[i]}")
    slots = np.concatenate([np.arange(offs[i], offs[i + 1]) for i in mine]) if mine else np.zeros(0, np.int64)
    vals = np.concatenate(sliding) if sliding else np.zeros(0, np.float32)
    if world > 1:
        all_slide = D.gather_posteriors(vals, slots, int(offs[-1]), device=comm_device)
        all_one = D.gather_posteriors(p_one, mine, n, device=comm_device)
    else:
        all_slide = np.zeros(int(offs[-1]), np.float32)
        all_slide[slots] = vals
        all_one = np.zeros(n, np.float32)
        all_one[mine] = p_one
    if rank != 0:
        return None
    # a clip shorter than the window yields no posterior (the reference's np.max would raise): 0
    pos = np.array([all_slide[offs[i]:offs[i + 1]].max() if offs[i + 1] > offs[i] else 0.0
                    for i in range(n) if labels[i]], np.float32)
    neg = (np.concatenate([all_slide[offs[i]:offs[i + 1]] for i in range(n) if not labels[i]])
           if (~labels).any() else np.zeros(0, np.float32))
    hours = sum(len(clips[i]) + 2 * CLIP_PAD for i in range(n) if not labels[i]) / 16000.0 / 3600.0
    thr, frr, fa, cnt = far_frr(pos, neg, max(int(labels.sum()), 1), hours, thresholds, windowsize, engine=engine)
    return {"thresholds": thr, "frr": frr, "fa_per_hour": fa, "fa_count": cnt, "frr_at_0.5_fa_per_hour": frr_at_fa(frr, fa, 0.5),
            "one_window_posteriors": all_one, "one_window_accuracy": float(((all_one >= 0.5) == labels).mean()),
            "positives": pos, "negatives": neg, "hours": hours, "sliding": all_slide, "sliding_offsets": offs,
            "windows": int(offs[-1]) + n, "posterior_checksum": float(all_slide.sum(dtype=np.float64))}


# ----------------------------------------------------------------------------------------------
# The evaluator script's own helpers (utils/evaluate_models.py:138-181, 183-253, 281-326), so that
# tools/evaluate_models.py is the reference's command line on the HIP path.
# ----------------------------------------------------------------------------------------------
def testset_files(base_path: str):
    """``test.json`` of a Hey-Snips style directory -> (wakeword wavs, other wavs) (``:138-147``)."""
    import json
    with open(base_path + "test.json", "r") as f:
        test_data = json.load(f)
    wake = [base_path + p["audio_file_path"] for p in test_data if p["is_hotword"]]
    other = [base_path + p["audio_file_path"] for p in test_data if not p["is_hotword"]]
    return wake, other


def concatenate_FA(wav_paths: Sequence[str], num_files: int, FAR_path: str, sample_rate: int = 16000) -> None:
    """One long negative wav: the first ``num_files`` clips joined by 100 ms of silence (``:150-160``;
    the reference uses pydub, here the PCM16 payloads are concatenated directly)."""
    gap = np.zeros(sample_rate // 10, np.int16)
    parts: List[np.ndarray] = []
    for i, path in enumerate(wav_paths[:max(num_files, 1)]):
        with wave.open(path, "rb") as w:
            if w.getsampwidth() != 2 or w.getnchannels() != 1 or w.getframerate() != sample_rate:
                raise ValueError(f"{path}: expected mono PCM16 at {sample_rate} Hz")
            pcm = np.frombuffer(w.readframes(w.getnframes()), np.int16)
        if i:
            parts.append(gap)
        parts.append(pcm)
    with wave.open(str(FAR_path), "wb") as w:
        w.setnchannels(1)
        w.setsampwidth(2)
        w.setframerate(sample_rate)
        w.writeframes(np.concatenate(parts).tobytes() if parts else b"")


def duration_test(FAR_path: str, sample_rate: int) -> float:
    """Duration in seconds of the negative evaluation wav (``:178-181``)."""
    return len(read_wav(str(FAR_path), sample_rate)) / sample_rate


def load_posteriors(models_dir, model_type, frame_width, sample_rate, eval_type, input_path, out_path, examine_audio=False,
                    rank: int = 0, world: int = 1, comm_device: Optional[str] = None, device: int = 0):
    """Pickle-cached posteriors (``:163-175``), computed by :func:`get_posterior_sharded` (``world`` = 1: one GPU, the
    same windows as :func:`get_posterior`); with ``world > 1`` every rank computes its share and rank 0 writes the cache."""
    import os
    import pickle
    if os.path.exists(str(out_path)):
        with open(str(out_path), "rb") as f:
            posteriors = pickle.load(f)
    else:
        posteriors = get_posterior_sharded(models_dir, model_type, eval_type, input_path, frame_width, sample_rate, rank,
                                           world, comm_device, device=device)
        if rank == 0:
            with open(str(out_path), "wb") as f:
                pickle.dump(posteriors, f)
    return np.squeeze(np.array(posteriors))


def plot_FRR_FAR(keyword_posteriors, no_keyword_posteriors, num_wakewords, total_duration_hrs, model_type,
                 models_dir: Optional[str] = None, show: bool = False) -> dict:
    """The curves ``plot_FRR_FAR`` draws (``:183-253``), returned as arrays (+ the BASELINE metric); they are
    drawn with matplotlib only when ``show`` is set and matplotlib is importable."""
    thr, frr, fa, cnt = far_frr(np.atleast_1d(keyword_posteriors), np.atleast_1d(no_keyword_posteriors), num_wakewords,
                                total_duration_hrs, models_dir=models_dir)
    out = {"model_type": model_type, "thresholds": thr, "FRR": frr, "FAR": fa, "FA_count": cnt,
           "frr_at_0.5_fa_per_hour": frr_at_fa(frr, fa, 0.5)}
    if show:  # pragma: no cover - interactive
        try:
            import matplotlib.pyplot as plt
        except ImportError:
            return out
        for x, y, xl, yl in ((thr, frr, "Posterior Threshold", "False Rejection Rate"),
                             (thr, fa, "Posterior Threshold", "False Accepts per Hour"),
                             (fa, frr, "False Alarms per Hour", "False Rejection Rate")):
            fig, ax = plt.subplots(1, 1)
            ax.set_facecolor("lightgray")
            plt.plot(x, y, label=model_type)
            plt.xlabel(xl)
            plt.ylabel(yl)
            plt.grid(color="white")
            plt.legend()
            plt.tight_layout()
            plt.show()
            plt.close()
    return out
