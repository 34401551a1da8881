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
CLIP_PAD = 8000  # 0.5 s of zeros each side of a clip (evaluate_models.py:52-53)
CLIP_HOP = 2     # mel rows between windows (evaluate_models.py:42)
_PIN = None  # page-locked staging buffer of clip_posteriors (torch tensor, grown on demand)


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
        per_file, starts = frame_schedule(self.padded.tolist(), 160, frame_length, carry_over)
        self.pos = np.asarray(starts, np.int64)
        self.n_frames = np.array([int(f.sum()) for f in per_file], np.int64)
        # first global frame credited to the file: the ring emits frames in order, so it is the count emitted before it
        self.F = np.array([((p - WINDOW) // 160 + 1 if p >= WINDOW else 0) for p in self.pos], np.int64)
        # (window i of a file starts at row hop * i of its frame list: the schedule drops `hop` rows per inference)
        self.n_win = np.array([len(window_schedule(fpc, self.T, self.hop)) for fpc in per_file], np.int64)
        self.offs = np.concatenate(([0], np.cumsum(self.n_win)))

    @property
    def total(self) -> int:
        return int(self.offs[-1])

    def shares(self, eval_type: str, world: int) -> List[List[Tuple[int, int, int]]]:
        """Per rank: ``(file, i0, i1)`` runs of windows.  Positives: whole files dealt longest-first round-robin
        (``dist.shard_by_length``); the negative stream: contiguous posterior ranges (``dist.split_stream``), cut
        where a range crosses a file boundary."""
        from . import dist as D
        out: List[List[Tuple[int, int, int]]] = [[] for _ in range(world)]
        if eval_type == "false_negatives":
            for r, files in enumerate(D.shard_by_length(self.lengths.tolist(), world)):
                out[r] = [(k, 0, int(self.n_win[k])) for k in sorted(files) if self.n_win[k] > 0]
        else:
            for r, (lo, hi) in enumerate(D.split_stream(self.total, world)):
                k = int(np.searchsorted(self.offs, lo, side="right")) - 1
                while lo < hi:
                    end = min(hi, int(self.offs[k + 1]))
                    if end > lo:
                        out[r].append((k, lo - int(self.offs[k]), end - int(self.offs[k])))
                    lo, k = end, k + 1
        return out

    def sample_range(self, k: int, i0: int, i1: int) -> Tuple[int, int]:
        """Stream samples the windows ``i0..i1-1`` of file ``k`` are functions of (with pre-emphasis 0, as ``Filter``'s
        default: no sample before the first frame is needed)."""
        g0 = int(self.F[k]) + self.hop * i0
        g1 = int(self.F[k]) + self.hop * (i1 - 1) + self.T - 1
        return 160 * g0, 160 * g1 + WINDOW  # (without carry every file has its own stream: pos = 0, F = 0)


def _stream_slice(plan: StreamPlan, load: Callable[[int], np.ndarray], cache: dict, s0: int, s1: int, k_hint: int) -> np.ndarray:
    """Samples ``[s0, s1)`` of the concatenated padded stream, touching only the files that overlap it."""
    pos = plan.pos if plan.carry else np.zeros_like(plan.pos)
    ks = [k_hint] if not plan.carry else [k for k in range(max(k_hint - 1, 0), len(pos)) if pos[k] < s1 and pos[k] + plan.padded[k] > s0]
    dt = np.int16
    for k in ks:
        if k not in cache:
            cache[k] = np.asarray(load(k))
        if cache[k].dtype != np.int16:
            dt = np.float32
    out = np.zeros(s1 - s0, dt)
    for k in ks:
        x = cache[k].astype(dt, copy=False)
        a = int(pos[k]) + plan.pad  # stream sample of x[0]
        lo, hi = max(s0, a), min(s1, a + len(x))
        if hi > lo:
            out[lo - s0: hi - s0] = x[lo - a: hi - a]
    return out


def _pieces_forward(eng: Engine, pieces: List[np.ndarray], n_win: List[int], hop: int, precise: bool = True) -> np.ndarray:
    """One front-end launch over ``pieces`` (each its own framing grid, frame j = samples [160 j, 160 j + 512)) and one
    model launch over their windows (piece p: ``n_win[p]`` windows at rows ``hop i``): detect rows, piece by piece."""
    import torch  # only to hold the device buffers of the batched launch

    n = len(pieces)
    if n == 0 or sum(n_win) == 0:
        return np.zeros((0, eng.n_out), np.float32)
    dev = torch.device("cuda", eng.ctx.device)
    nf = np.array([eng.num_frames(len(p)) for p in pieces], np.int64)
    foffs = np.concatenate(([0], np.cumsum(nf)))
    total_f = int(foffs[-1])
    for p, f, w in zip(pieces, nf, n_win):
        assert w == 0 or (w - 1) * hop + eng.window <= f, "piece too short for its windows"
    d_mel = torch.empty((max(total_f, 1), eng.n_mel), dtype=torch.float32, device=dev)
    if all(p.dtype == np.int16 for p in pieces):
        # librosa's floats are int16 / 32768 exactly: the device front end divides (correctly rounded) by the same 32768
        soffs = np.concatenate(([0], np.cumsum([len(p) for p in pieces]))).astype(np.int64)
        pcm = np.zeros(int(soffs[-1]) + 16, np.int16)  # (the kernel's vector loads may run a few samples past the end)
        for i, p in enumerate(pieces):
            pcm[soffs[i]: soffs[i + 1]] = p
        d_pcm, d_so, d_fo = torch.from_numpy(pcm).to(dev), torch.from_numpy(soffs).to(dev), torch.from_numpy(foffs).to(dev)
        torch.cuda.synchronize(dev)
        eng.logmel_dev(d_pcm.data_ptr(), d_so.data_ptr(), d_fo.data_ptr(), n, total_f, int(nf.max()), d_mel.data_ptr(),
                       frontend_params(32768.0, False, 0.0, 160, precise))
    else:
        mels = eng.logmel([np.asarray(p, np.float32) for p in pieces], frontend_params(1.0, False, 0.0, 160, precise))
        d_mel[:total_f] = torch.from_numpy(np.concatenate(mels)).to(dev)
    n_tot = int(sum(n_win))
    d_out = torch.empty((n_tot, eng.n_out), dtype=torch.float32, device=dev)
    torch.cuda.synchronize(dev)
    # the library picks the tail kernel by the share's window count; every form of the CRNN associates its sums the same way
    # (csrc/crnn.hip: gru_step), so a rank's posteriors do not depend on how many windows its launch holds
    eng.forward_segments_dev(d_mel.data_ptr(), total_f, foffs[:-1], np.asarray(n_win, np.int32), hop, d_out.data_ptr())
    eng.ctx.synchronize()
    return d_out.cpu().numpy()


def get_posterior_sharded(models_dir, model_type, eval_type, test_files, frame_width, sample_rate, rank: int = 0,
                          world: int = 1, comm_device: Optional[str] = None,
                          loader: Optional[Callable[[str], np.ndarray]] = None, lengths: Optional[Sequence[int]] = None,
                          carry_over: bool = True, device: int = 0, engine: Optional[Engine] = None,
                          precise: bool = True) -> list:
    """:func:`get_posterior` with its windows dealt to ``world`` ranks (one process per GPU; ``torch.distributed``
    initialised by the caller when ``world > 1``).  ``"false_negatives"``: whole files, longest first round-robin;
    ``"false_accepts"``: the window list - for the reference's evaluator ONE long wav (``evaluate_models.py:317-321``) -
    cut into ``world`` contiguous posterior ranges (``dist.split_stream``).  A rank loads and front-ends only the samples
    its windows are functions of: posterior ``i`` of a file needs the global frames ``[F + 2 i, F + 2 i + T)``, i.e. each
    range re-reads a ``T - 2``-frame overlap and the results are exact.  The one exchange is the posterior gather.  Every
    rank returns the full list :func:`get_posterior` returns.

    ``test_files``: paths (default loader :func:`read_wav_pcm`; ``lengths`` default = wav headers) or arrays already in
    memory (int16 PCM, or float32 samples in [-1, 1)).  ``precise=False``: the fp32-FFT front end (``ww_frontend_params.precise``
    = 0) instead of the reference's float64 STFT."""
    if model_type not in ("CRNN", "Wavenet"):
        raise ValueError("model_type must be 'CRNN' or 'Wavenet'")
    if eval_type not in ("false_negatives", "false_accepts"):
        raise ValueError("eval_type must be 'false_negatives' or 'false_accepts'")
    eng: Engine = engine or engine_for(models_dir, device)
    frame_length = sample_rate // 1000 * frame_width
    in_memory = len(test_files) > 0 and not isinstance(test_files[0], (str, bytes)) and not hasattr(test_files[0], "__fspath__")
    if in_memory:
        load = lambda k: np.asarray(test_files[k])  # noqa: E731
        lengths = [len(x) for x in test_files] if lengths is None else lengths
    else:
        rd = loader or (lambda p: read_wav_pcm(p, sample_rate))
        load = lambda k: rd(str(test_files[k]))  # noqa: E731
        if lengths is None:
            lengths = [wav_length(str(f), sample_rate) for f in test_files] if loader is None else [len(rd(str(f))) for f in test_files]
    if len(test_files) == 0:
        return []
    plan = StreamPlan(lengths, eng.window, frame_length, sample_rate, 2, carry_over)
    mine = plan.shares(eval_type, world)[rank]
    cache: dict = {}
    pieces, n_win, slots = [], [], []
    for k, i0, i1 in mine:
        s0, s1 = plan.sample_range(k, i0, i1)
        pieces.append(_stream_slice(plan, load, cache, s0, s1, k))
        n_win.append(i1 - i0)
        slots.append(np.arange(plan.offs[k] + i0, plan.offs[k] + i1))
        for j in [j for j in cache if j < k - 1]:
            del cache[j]
    vals = _pieces_forward(eng, pieces, n_win, plan.hop, precise)[:, eng.posterior_index]
    slots = np.concatenate(slots) if slots else np.zeros(0, np.int64)
    if world > 1:
        from . import dist as D
        post = D.gather_posteriors(vals, slots, plan.total, device=comm_device)
    else:
        post = np.zeros(plan.total, np.float32)
        post[slots] = vals
    if eval_type == "false_negatives":
        return [np.max(post[plan.offs[k]:plan.offs[k + 1]]) for k in range(len(plan.n_win))]  # raises on an empty clip, like the reference
    return post.tolist()


def join_negatives(clips: Sequence[np.ndarray], num_files: int, sample_rate: int = 16000) -> np.ndarray:
    """``concatenate_FA`` (``evaluate_models.py:150-160``) on PCM in memory: ``clips[0] + (100 ms silence + clip) for
    clips[1:num_files]``."""
    gap = np.zeros(sample_rate // 10, np.int16)
    parts = [np.asarray(clips[0], np.int16)]
    for c in clips[1:max(num_files, 0)]:
        parts += [gap, np.asarray(c, np.int16)]
    return np.concatenate(parts)


def evaluate_negative_stream_sharded(engine: Engine, stream_pcm: np.ndarray, rank: int = 0, world: int = 1,
                                     comm_device: Optional[str] = None, precise: bool = True) -> np.ndarray:
    """The reference's false-accept leg (``evaluate_models.py:317-321``: ``get_posterior(..., "false_accepts",
    [FAR_path])``) on one long PCM stream, cut into ``world`` contiguous posterior ranges; full posterior array on
    every rank (smoothing across the cuts happens after the gather, :func:`far_frr`)."""
    return np.asarray(get_posterior_sharded(engine.model_dir, "CRNN" if engine.is_crnn else "Wavenet", "false_accepts",
                                            [np.asarray(stream_pcm)], 20, 16000, rank, world, comm_device, engine=engine,
                                            precise=precise), np.float32)


def evaluate_reference_flow_sharded(engine: Engine, clips: Sequence[np.ndarray], labels: Sequence[int], rank: int = 0,
                                    world: int = 1, comm_device: Optional[str] = None, thresholds=None, windowsize: int = 30,
                                    precise: bool = True):
    """``utils/evaluate_models.py`` ``main()`` (``:281-326``) on labelled int16 clips in memory, sharded over ``world``
    ranks: the wake-word clips go file by file through one never-reset ``Filter`` (quirk C2; utterance-sharded, a rank
    re-reads the <= 511-sample tail of a file's predecessor), the first ``num_wakewords`` other clips are joined by 100 ms
    of silence into ONE stream (``concatenate_FA``) that is evaluated continuously and cut into contiguous posterior ranges
    (:func:`evaluate_negative_stream_sharded`), hours = the joined stream's duration; rank 0 smooths and sweeps.
    Returns the result dict on rank 0 and ``None`` elsewhere."""
    labels = np.asarray(labels).astype(bool)
    wake = [np.asarray(c, np.int16) for c, l in zip(clips, labels) if l]
    other = [np.asarray(c, np.int16) for c, l in zip(clips, labels) if not l]
    mtype = "CRNN" if engine.is_crnn else "Wavenet"
    num_wakewords = len(wake)
    pos = np.asarray(get_posterior_sharded(engine.model_dir, mtype, "false_negatives", wake, 20, 16000, rank, world,
                                           comm_device, engine=engine, precise=precise), np.float32)
    stream = join_negatives(other, num_wakewords) if other else np.zeros(0, np.int16)
    neg = (evaluate_negative_stream_sharded(engine, stream, rank, world, comm_device, precise) if len(stream)
           else np.zeros(0, np.float32))
    if rank != 0:
        return None
    hours = len(stream) / 16000.0 / 3600.0
    thr, frr, fa, cnt = far_frr(pos, neg, max(num_wakewords, 1), hours, thresholds, windowsize, engine=engine)
    return {"thresholds": thr, "frr": frr, "fa_per_hour": fa, "fa_count": cnt, "frr_at_0.5_fa_per_hour": frr_at_fa(frr, fa, 0.5),
            "positives": pos, "negatives": neg, "hours": hours, "num_wakewords": num_wakewords,
            "negative_clips_joined": min(len(other), max(num_wakewords, 1)), "windows": int(len(neg)) + int(num_wakewords),
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
