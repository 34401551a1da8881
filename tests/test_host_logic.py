"""Host-side mirror of the reference interface (CPU only): RingBuffer, SpeechContext,
SpeechPipeline, ActivationTimeout, frame/window schedules - against fixtures recorded from the
reference's own classes (tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import pytest

from wwhip.activation_timeout import ActivationTimeout
from wwhip.context import SpeechContext
from wwhip.pipeline import SpeechPipeline
from wwhip.ring_buffer import RingBuffer
from wwhip import evaluate as E
from oracle import numpy_ref as NR


def test_ring_buffer_replays_reference_trace(golden):
    script = json.load(open(os.path.join(golden, "ringbuffer_trace.json")))
    for case in script:
        shape = case["shape"]
        rb = RingBuffer(shape=list(shape))
        for rec in case["ops"]:
            op = rec["op"]
            raised = None
            ret = None
            try:
                if op == "write":
                    v = rec["arg"]
                    rb.write(np.full(shape[1:], v, np.float32) if len(shape) > 1 else np.float32(v))
                elif op == "read":
                    ret = np.asarray(rb.read()).ravel().tolist()
                elif op == "rewind":
                    assert rb.rewind() is rb
                elif op == "seek":
                    assert rb.seek(rec["arg"]) is rb
                elif op == "reset":
                    assert rb.reset() is rb
                elif op == "fill":
                    assert rb.fill(rec["arg"]) is rb
                elif op == "read_all":
                    out = rb.read_all()
                    ret = np.asarray(out).ravel().tolist()
                    assert list(out.shape) == rec["ret_shape"]
            except IndexError as e:
                raised = str(e)
            assert raised == rec.get("raises"), (op, rec)
            if "ret" in rec and op == "read":
                assert ret == rec["ret"]
            if "ret" in rec and op == "read_all":
                # slots never written hold np.empty garbage in both implementations: compare the
                # entries the reference run had defined (finite, small) only when identical length
                assert len(ret) == len(rec["ret"])
            st = rec["state"]
            assert (rb.is_empty, rb.is_full, rb.capacity) == (st["empty"], st["full"], st["capacity"]), (op, rec)


def test_ring_buffer_values_and_block_write():
    rb = RingBuffer([5])
    rb.write_block(np.arange(4, dtype=np.float32))
    assert len(rb) == 4 and not rb.is_full
    rb.write(np.float32(4))
    assert rb.is_full
    with pytest.raises(IndexError, match="Buffer is full"):
        rb.write(np.float32(9))
    assert rb.read().tolist() == [0.0] and rb.read().shape == (1,)
    rb.write_block(np.array([5, 6], np.float32))
    assert rb.read_all().tolist() == [2.0, 3.0, 4.0, 5.0, 6.0]
    assert rb.is_empty
    with pytest.raises(IndexError, match="Buffer is empty"):
        rb.read()
    rb.fill(0.0)
    assert rb.is_full and rb.read_all().tolist() == [0.0] * 5
    rb2 = RingBuffer([3, 2])
    rb2.fill(1.0).rewind().seek(1)
    rb2.write(np.array([7, 8], np.float32))
    np.testing.assert_array_equal(rb2.read_all(), [[1, 1], [1, 1], [7, 8]])


def test_pipeline_and_context_replay_reference_trace(golden):
    want = json.load(open(os.path.join(golden, "pipeline_trace.json")))
    trace = []

    class Src:
        n = 0

        def start(self): trace.append("src.start")
        def stop(self): trace.append("src.stop")
        def close(self): trace.append("src.close")

        def read(self):
            self.n += 1
            trace.append(f"src.read{self.n}")
            return np.zeros(320, np.int16)

    class Stage:
        def __init__(self, name, ref):
            self.name, self.ref = name, ref

        def __call__(self, ctx, frame):
            trace.append(f"{self.name}.call active={ctx.is_active} speech={ctx.is_speech}")
            n = self.ref["n"]
            if self.name == "a" and n == 1: ctx.is_speech = True
            if self.name == "b" and n == 2: ctx.is_active = True
            if self.name == "b" and n == 3: ctx.is_active = False
            if self.name == "b" and n == 4: self.ref["pipe"].stop()

        def close(self): trace.append(f"{self.name}.close")

    ref = {"n": 0}
    pipe = SpeechPipeline(Src(), [Stage("a", ref), Stage("b", ref)])
    ref["pipe"] = pipe

    @pipe.event
    def on_activate(ctx): trace.append("event.activate")

    @pipe.event
    def on_deactivate(ctx): trace.append("event.deactivate")

    @pipe.event(name="step")
    def counter(ctx):
        ref["n"] += 1
        trace.append(f"event.step{ref['n']}")

    pipe.start(); pipe.start(); pipe.run()
    trace.append(f"running={pipe.is_running}")
    ctx = SpeechContext()
    ctx.is_active = True; ctx.is_active = True; ctx.transcript = "x"; ctx.confidence = 0.5; ctx.reset()
    trace.append(f"ctx {ctx.is_active} {ctx.is_speech} {ctx.transcript!r} {ctx.confidence}")
    assert trace == want


def test_activation_timeout_semantics():
    ctx = SpeechContext()
    t = ActivationTimeout(frame_width=20, min_active=100, max_active=200)
    ctx.is_active = True
    ctx.is_speech = True
    for _ in range(5):
        t(ctx)
    assert ctx.is_active            # 5 frames = min_active: not yet over
    ctx.is_speech = False
    t(ctx)                          # vad fall after min_active -> deactivate
    assert not ctx.is_active
    ctx.is_active = True
    for _ in range(11):
        t(ctx)
    assert not ctx.is_active        # max_active exceeded


def test_activation_timeout_replays_reference_trace(golden):
    """tests/golden/activation_timeout_trace.json was recorded from the reference's own ActivationTimeout and
    SpeechContext (spokestack/activation_timeout.py:25-38; tests/golden/make_golden.py): same script in, same
    is_active after every frame and the same activate / deactivate events out."""
    cases = json.load(open(os.path.join(golden, "activation_timeout_trace.json")))
    assert len(cases) == 15
    for case in cases:
        ctx = SpeechContext()
        events = []
        for name in ("activate", "deactivate"):
            ctx.add_handler(name, (lambda n: (lambda c: events.append(n)))(name))
        stage = ActivationTimeout(**case["kwargs"])
        got_events = []
        for t, (sp, act, rst, want) in enumerate(zip(case["is_speech"], case["activate"], case["reset"], case["is_active"])):
            ctx.is_speech = sp == "1"
            if act == "1":
                ctx.is_active = True
            if rst == "1":
                stage.reset()
            n_ev = len(events)
            stage(ctx, None)
            assert ctx.is_active == (want == "1"), (case["kwargs"], t)
            got_events += [[t, e] for e in events[n_ev:]]
        assert got_events == case["events"]


def test_frame_schedule_matches_reference_ring(golden):
    z = np.load(os.path.join(golden, "framing.npz"))
    for name in ["single_24000", "single_short", "single_tiny", "two_files_3200", "ragged"]:
        files = z[name + ".files"]
        padded = [int(-(-n // 320) * 320) for n in files]
        per_file, starts = E.frame_schedule(padded, 160, 320, carry_over=True)
        assert [int(p.sum()) for p in per_file] == z[name + ".frames_per_file"].tolist()
        assert np.concatenate(per_file).tolist() == z[name + ".frames_per_chunk"].tolist()
        total = int(sum(p.sum() for p in per_file))
        assert (160 * np.arange(total)).tolist() == z[name + ".starts"].tolist()


def test_window_schedule_matches_reference_loop():
    rng = np.random.default_rng(0)
    for T in (151, 182):
        for n in (100, 48000, 30001):
            x = rng.normal(0, 0.1, n).astype(np.float32)
            filt = NR.RefFilter(lambda a: np.zeros((1, 40), np.float32))
            filt.stft_mag = lambda: np.zeros(257, np.float32)      # numerics irrelevant here
            seen = []
            NR.sliding_posteriors(filt, x, T, lambda w: seen.append(1) or 0.0)
            padded = len(x) + 16000
            padded += (-padded) % 320
            fpc, _ = E.frame_schedule([padded])
            ws = E.window_schedule(fpc[0], T, 2)
            assert len(ws) == len(seen)
            assert ws.tolist() == (2 * np.arange(len(ws))).tolist()


def test_stream_plan_matches_reference_loop_over_files():
    """wwhip.evaluate.StreamPlan (the arithmetic every rank of the sharded evaluation shares) against the literal per-file
    loop of utils/evaluate_models.py:45-88 driven by ONE never-reset Filter (quirk C2): the "mel row" of global frame j is
    filled with j, so every window reports which global frames it was made of."""
    rng = np.random.default_rng(3)
    for T in (151, 182):
        lens = [30000, 41000, 4000, 24000, 51234]
        counter = [0]

        def filter_model(mag):
            counter[0] += 1
            return np.full((1, 40), counter[0] - 1, np.float32)

        filt = NR.RefFilter(filter_model)
        filt.stft_mag = lambda: np.zeros(257, np.float32)
        per_file = []
        for n in lens:
            wins = []
            NR.sliding_posteriors(filt, rng.normal(0, 0.1, n).astype(np.float32), T,
                                  lambda w: wins.append((int(w[0, 0]), int(w[-1, 0]), len(w))) or 0.0)
            per_file.append(wins)
        plan = E.StreamPlan(lens, T)
        assert plan.n_win.tolist() == [len(w) for w in per_file]
        assert plan.n_win[2] == 0                                   # a 0.25 s clip never fills a window
        for k, wins in enumerate(per_file):
            for i, (first, last, rows) in enumerate(wins):
                assert rows == T and first == plan.F[k] + 2 * i and last == first + T - 1
                s0, s1 = plan.sample_range(k, i, i + 1)
                assert s0 == 160 * first and s1 == 160 * last + 512
                assert s0 >= plan.pos[k] - 511 and s1 <= plan.pos[k] + plan.padded[k]   # at most the predecessor's tail
        # every window lands in exactly one rank's share, whatever the world size
        for world in (1, 2, 3, 5, 8):
            for eval_type in ("false_negatives", "false_accepts"):
                seen = np.zeros(plan.total, int)
                for share in plan.shares(eval_type, world):
                    for k, i0, i1 in share:
                        assert 0 <= i0 < i1 <= plan.n_win[k]
                        seen[plan.offs[k] + i0: plan.offs[k] + i1] += 1
                assert (seen == 1).all()
            sizes = [sum(i1 - i0 for _, i0, i1 in sh) for sh in plan.shares("false_accepts", world)]
            assert max(sizes) - min(sizes) <= 1                     # contiguous ranges of equal size (dist.split_stream)


def test_shares_and_chunk_cuts_as_arrays_equal_the_per_file_loops():
    """Round 5: every rank works out every rank's share and cuts its own into chunks WITHOUT a Python-level pass over the files
    (`StreamPlan.shares_arr`, `_PosteriorJob._cut`: the part of an evaluation call that no rank count divides).  Against the
    loops they replace, restated here: the longest-first round-robin deal / the contiguous posterior ranges walked file by
    file, and the run-by-run chunk grouping - over random file lists incl. files without a window, several world sizes and
    first-chunk sizes."""
    from wwhip import dist as D
    rng = np.random.default_rng(31)

    def shares_loop(plan, eval_type, world):
        out = [[] for _ in range(world)]
        if eval_type == "false_negatives":
            nw = plan.n_win.tolist()
            for r, files in enumerate(D.shard_by_length(plan.lengths.tolist(), world)):
                out[r] = [(k, 0, nw[k]) for k in sorted(files) if nw[k] > 0]
        else:
            for r, (lo, hi) in enumerate(D.split_stream(plan.total, world)):
                k = int(np.searchsorted(plan.offs, lo, side="right")) - 1
                while lo < hi:
                    end = min(hi, int(plan.offs[k + 1]))
                    if end > lo:
                        out[r].append((k, lo - int(plan.offs[k]), end - int(plan.offs[k])))
                    lo, k = end, k + 1
        return out

    def cut_loop(plan, runs, first, cut=True):
        # (cut=False: the wake-word clips' share - one value per run, the clip's maximum picked on the device - keeps every run
        # whole; a clip longer than a chunk is a chunk of its own)
        per_win = 160 * plan.hop
        out, cur, cur_n = [], [], 0
        size = max(min(first, E._CHUNK_SAMPLES), per_win)
        for k, i0, i1 in runs:
            while cut and i1 - i0 > (size + size // 2) // per_win:
                if cur:
                    out.append(cur); cur, cur_n = [], 0; size = min(2 * size, E._CHUNK_SAMPLES)
                w = max(size // per_win, 1)
                out.append([(k, i0, i0 + w)]); size = min(2 * size, E._CHUNK_SAMPLES)
                i0 += w
            n = 160 * (plan.hop * (i1 - i0 - 1) + plan.T - 1) + 512
            if cur and cur_n + n > size:
                out.append(cur); cur, cur_n = [], 0; size = min(2 * size, E._CHUNK_SAMPLES)
            cur.append((k, i0, i1))
            cur_n += n
        if cur:
            out.append(cur)
        return out

    class Eng:
        window = 151

    for trial in range(25):
        n = int(rng.integers(1, 400))
        lens = rng.integers(200, 50000, n).tolist()
        if trial % 3 == 0:
            lens[int(rng.integers(0, n))] = 3_000_000        # one long file: cut into window ranges
        plan = E.StreamPlan(lens, 151)
        for world in (1, 2, 3, 8):
            for et in ("false_negatives", "false_accepts"):
                want = shares_loop(plan, et, world)
                assert plan.shares(et, world) == want
                got = plan.shares_arr(et, world)
                assert all(a.dtype == np.int64 and a.shape == (len(w), 3) for a, w in zip(got, want))
                for first in (1 << 14, 1 << 18, 1 << 21):
                    for rank in {0, world - 1}:
                        job = E._PosteriorJob(Eng, et, [np.zeros(1, np.int16)] * n, 20, 16000, rank, world, None, lens, True,
                                              E._Phases(None), None, first_chunk=first)
                        cuts = [[tuple(r) for r in c.runs.tolist()] for c in job.chunks]
                        assert cuts == cut_loop(plan, want[rank], first, cut=et == "false_accepts"), (trial, world, et, first)
                        sl = job.slots_of(job.mine)
                        idx = np.arange(plan.total)[sl] if isinstance(sl, slice) else sl
                        assert idx.tolist() == [plan.offs[k] + i for k, i0, i1 in want[rank] for i in range(i0, i1)]


def test_stream_plan_closed_form_equals_the_per_file_schedules():
    """StreamPlan computes the layout of a whole test split at once when a chunk is exactly `hop` mel hops (the reference's
    320-sample chunks and hop 2): the same pos / F / n_frames / n_win as the per-file frame_schedule + window_schedule (which the
    test above pins on the reference's loop), over random file lists incl. empty, sub-window and single-chunk files, both
    carry modes, three window lengths."""
    rng = np.random.default_rng(0)
    for trial in range(200):
        n = int(rng.integers(1, 12))
        lens = rng.integers(0, 60000, n)
        if trial % 5 == 0:
            lens[rng.integers(0, n)] = 0
        if trial % 7 == 0:
            lens[:] = rng.integers(0, 700, n)
        for T in (151, 182, 40):
            for carry in (True, False):
                fast = E.StreamPlan(lens, T, 320, 16000, 2, carry)
                slow = E.StreamPlan(lens, T, 320, 16000, 2, carry)
                slow._per_file()
                for f in ("pos", "F", "n_frames", "n_win"):
                    assert np.array_equal(getattr(fast, f), getattr(slow, f)), (trial, T, carry, f, lens.tolist())


def test_joined_pcm_is_concatenate_fa_without_the_copy():
    """JoinedPCM (the negative stream of evaluate_models.py:150-160 as a view on its clips) against join_negatives: length,
    materialised array, and every (array, offset, count, position) run of arbitrary sample ranges."""
    rng = np.random.default_rng(1)
    clips = [rng.integers(-3000, 3000, int(n)).astype(np.int16) for n in rng.integers(1, 5000, 9)]
    for num in (1, 4, 9, 20):
        want = E.join_negatives(clips, num)
        j = E.join_negatives_lazy(clips, num)
        assert len(j) == len(want) and j.dtype == np.int16
        np.testing.assert_array_equal(j.to_array(), want)
        for _ in range(50):
            a = int(rng.integers(0, len(want)))
            n = int(rng.integers(1, len(want) - a + 1))
            got = np.zeros(n, np.int16)
            for arr, aa, nn, o in j.runs(a, n):
                got[o:o + nn] = arr[aa:aa + nn]
            np.testing.assert_array_equal(got, want[a:a + n])
    bad = E.JoinedPCM([np.zeros(4, np.int16), np.zeros(4, np.float32)], 1600)   # a clip is looked at where it is first used
    assert len(bad) == 1608 and bad.part(0).dtype == np.int16
    for use in (lambda: bad.to_array(), lambda: bad.addresses(0, 2), lambda: list(bad.runs(0, 1608))):
        with pytest.raises(TypeError):
            use()
    lists = E.JoinedPCM([[1, 2, 3], np.arange(10, dtype=np.int32)[::2]], 2)      # integer clips of another layout: converted once
    np.testing.assert_array_equal(lists.to_array(), [1, 2, 3, 0, 0, 0, 2, 4, 6, 8])
    assert lists.addresses(0, 2).all()


@pytest.mark.parametrize("frame_width", [10, 20, 25, 40])
def test_stream_plan_other_chunk_sizes_and_no_carry(frame_width):
    """The same pin for other `frame_width`s of the evaluator's command line (10 ms: one mel frame per chunk, so the one-inference-
    per-chunk rule lets the buffer grow; 25 ms = 400 samples: not a multiple of the hop, frames per chunk alternate 2, 3) and for
    `carry_over=False` (a fresh ring per file)."""
    rng = np.random.default_rng(5)
    fl = 16 * frame_width
    lens = [30000, 12345, 41000]
    for carry in (True, False):
        counter = [0]

        def filter_model(mag):
            counter[0] += 1
            return np.full((1, 40), counter[0] - 1, np.float32)

        filt = NR.RefFilter(filter_model)
        filt.stft_mag = lambda: np.zeros(257, np.float32)
        plan = E.StreamPlan(lens, 151, fl, carry_over=carry)
        for k, n in enumerate(lens):
            if not carry:                                           # a fresh Filter per file, frame ids restart at 0
                counter[0] = 0
                filt = NR.RefFilter(filter_model)
                filt.stft_mag = lambda: np.zeros(257, np.float32)
            wins = []
            NR.sliding_posteriors(filt, rng.normal(0, 0.1, n).astype(np.float32), 151,
                                  lambda w: wins.append((int(w[0, 0]), int(w[-1, 0]))) or 0.0, frame_length=fl)
            assert plan.n_win[k] == len(wins), (frame_width, carry, k)
            for i, (first, last) in enumerate(wins):
                assert first == plan.F[k] + 2 * i and last == first + 150
                s0, s1 = plan.sample_range(k, i, i + 1)
                base = 0 if carry else 0
                assert s0 == base + 160 * first and s1 == base + 160 * last + 512
        if carry:
            assert (plan.pos[1:] > 0).all()
        else:
            assert (plan.pos == 0).all() and (plan.F == 0).all()


def test_join_negatives_is_concatenate_FA(tmp_path):
    """join_negatives (in memory) == concatenate_FA (wav files): first num_files clips, 100 ms of silence between."""
    import wave
    rng = np.random.default_rng(4)
    clips = [np.clip(rng.normal(0, 3000, n), -32768, 32767).astype(np.int16) for n in (5000, 7000, 3000, 9000)]
    paths = []
    for i, c in enumerate(clips):
        paths.append(str(tmp_path / f"n{i}.wav"))
        with wave.open(paths[-1], "wb") as w:
            w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(c.tobytes())
    for num in (3, 1, 0, 9):
        E.concatenate_FA(paths, num, str(tmp_path / "far.wav"))
        want = E.read_wav_pcm(str(tmp_path / "far.wav"))
        got = E.join_negatives(clips, num)
        np.testing.assert_array_equal(got, want)
        assert E.wav_length(str(tmp_path / "far.wav")) == len(got)
    z = np.zeros(1600, np.int16)
    np.testing.assert_array_equal(E.join_negatives(clips, 3), np.concatenate([clips[0], z, clips[1], z, clips[2]]))


def test_window_schedule_closed_form_equals_the_loop():
    """window_schedule (cumulative-maximum closed form) against the literal loop on random chunk patterns: chunks of 0..5 frames,
    windows of several lengths, hops 1..3 - including files that never fill a window and files that end mid-buffer."""
    rng = np.random.default_rng(12)
    for trial in range(300):
        K = int(rng.integers(0, 400))
        fpc = rng.integers(0, int(rng.integers(1, 6)) + 1, K)
        T = int(rng.choice([1, 7, 151, 182]))
        hop = int(rng.integers(1, 4))
        got = E.window_schedule(fpc, T, hop)
        want = E._window_schedule_loop(fpc, T, hop)
        assert got.tolist() == want.tolist(), (trial, K, T, hop)
    assert E.window_schedule(np.zeros(0, int), 151).size == 0


def test_read_wav_roundtrip(tmp_path):
    import wave
    p = str(tmp_path / "a.wav")
    pcm = (np.arange(-100, 100) * 100).astype(np.int16)
    with wave.open(p, "wb") as w:
        w.setnchannels(1); w.setsampwidth(2); w.setframerate(16000); w.writeframes(pcm.tobytes())
    x = E.read_wav(p)
    np.testing.assert_array_equal(x, pcm.astype(np.float32) / np.float32(32768))
    with pytest.raises(ValueError):
        E.read_wav(p, sample_rate=8000)


def test_load_data_truncates_and_pads():
    feats = [np.ones((200, 40), np.float32), np.ones((10, 40), np.float32) * 2]
    X, y = E.load_data(feats, [1, 0], 151, 40)
    assert X.shape == (2, 151, 40) and X[0].sum() == 151 * 40 and X[1, 10:].sum() == 0 and X[1, :10].sum() == 800
    np.testing.assert_array_equal(X, NR.load_h5_like(feats, 151, 40))
    assert y.tolist() == [1, 0]


def test_pcm_quotient_sequence_is_exact_for_all_int16():
    """The front-end kernels form int16/divisor as q0 = a*r, e = fma(-b, q0, a), q = fma(e, r, q0)
    (csrc/frontend.hip:pcm_quot).  Exhaustive proof with exact rational arithmetic that this equals
    the correctly rounded fp32 division the reference performs (tflite.py:150) for both divisors."""
    from fractions import Fraction
    f32 = np.float32

    def rn32(fr):
        y = f32(float(fr))
        cands = [np.nextafter(y, f32(-np.inf)), y, np.nextafter(y, f32(np.inf))]
        return f32(min(cands, key=lambda c: (abs(Fraction(float(c)) - fr), int(f32(c).view(np.uint32)) & 1)))

    for div in (32767.0, 32768.0):
        b = f32(div)
        r = f32(1.0) / b
        a = np.arange(-32768, 32768).astype(np.float32)
        q0 = a * r
        want = a / b
        for ai, qi, wi in zip(a[::1], q0, want):
            e = rn32(Fraction(float(ai)) - Fraction(float(b)) * Fraction(float(qi)))
            q1 = rn32(Fraction(float(qi)) + Fraction(float(e)) * Fraction(float(r)))
            assert q1 == wi, (div, ai)


def test_vad_hysteresis_and_trigger():
    """spokestack/vad/webrtc.py:52-77,100-110 with a scripted frame classifier."""
    from spokestack.vad.webrtc import VoiceActivityDetector, VoiceActivityTrigger
    from wwhip.context import SpeechContext
    from wwhip.vad import VadBank
    script = [0, 1, 1, 0, 1, 1, 1, 1, 0, 0, 1, 0, 0, 0, 0, 1]
    it = iter(script)
    vad = VoiceActivityDetector(frame_width=20, vad_rise_delay=60, vad_fall_delay=80, classifier=lambda b, sr: next(it))
    trig = VoiceActivityTrigger()
    ctx = SpeechContext()
    got, active = [], []
    frame = np.zeros(320, np.int16)
    for _ in script:
        vad(ctx, frame)
        trig(ctx, frame)
        got.append(bool(ctx.is_speech))
        active.append(bool(ctx.is_active))
    # rise needs a run of 3 speech frames (60 // 20), fall a run of 4 non-speech frames (80 // 20)
    want = [False, False, False, False, False, False, True, True, True, True, True, True, True, True, False, False]
    assert got == want
    assert active == [False] * 6 + [True] * 10           # activated on the rising edge, never cleared by the trigger
    # zero delays: is_speech follows the classifier (run_length >= 0 always holds)
    it2 = iter(script)
    vad0 = VoiceActivityDetector(classifier=lambda b, sr: next(it2))
    ctx0 = SpeechContext()
    out0 = []
    for _ in script:
        vad0(ctx0, frame)
        out0.append(bool(ctx0.is_speech))
    assert out0 == [bool(v) for v in script]
    # the batched form takes the same decisions, stream by stream
    rng = np.random.default_rng(3)
    raw = rng.random((200, 7)) < 0.6
    bank = VadBank(7, 20, 60, 80)
    singles = []
    for s in range(7):
        col = iter(raw[:, s])
        singles.append((VoiceActivityDetector(frame_width=20, vad_rise_delay=60, vad_fall_delay=80,
                                              classifier=lambda b, sr, col=col: next(col)), SpeechContext()))
    for t in range(200):
        sp = bank.step(raw[t])
        for s, (v, c) in enumerate(singles):
            v(c, frame)
            assert bool(c.is_speech) == bool(sp[s])
    try:
        import webrtcvad  # noqa: F401
    except ImportError:
        with pytest.raises(RuntimeError):
            VoiceActivityDetector()                      # no webrtcvad: the classifier must be given


@pytest.mark.timeout(120)
def test_staging_pipeline_hands_a_planning_failure_to_the_caller():
    """wwhip.evaluate._run_jobs: a failure while a chunk is being planned (the loader fails inside the first chunk; a job
    constructor raises) reaches the caller as itself and nothing was submitted or launched - the part of the pipeline's
    failure handling that needs no GPU (tests/test_gpu_bench_eval.py has a failing launch with uploads in flight)."""
    from wwhip import evaluate as E

    class Ctx:
        device = 0

        def synchronize(self):
            raise AssertionError("nothing was launched: nothing to wait for")

    class Eng:
        window, ctx = 151, Ctx()

    def loader(path):
        if int(path) >= 2:
            raise OSError(f"cannot read {path}")
        return np.zeros(20000, np.int16)

    ph = E._Phases(None)
    with pytest.raises(OSError, match="cannot read 2"):
        E._run_jobs(Eng, [lambda: E._PosteriorJob(Eng, "false_negatives", [str(i) for i in range(8)], 20, 16000, 0, 1, loader,
                                                   [20000] * 8, True, ph, None, first_chunk=1 << 30)], True, ph, None)
    with pytest.raises(ZeroDivisionError):
        E._run_jobs(Eng, [lambda: 1 // 0], True, ph, None)


# ---- the banked stages (BASELINE config 5 at the plugin surface): host logic only, a stub in place of the device bank -------

class _StubStreamBank:
    """Stands where ``StreamBank`` stands in ``WakewordBank``: 'delivers' scripted posteriors (an active stream is not sampled,
    a frame is analysed only while the stream's VAD bit is set - spokestack/wakeword/tflite.py:139-140,166), then runs the
    library's host pass ``ww_trigger_bank_step`` exactly as ``ww_stream_step_trigger`` does after the tick."""

    def __init__(self, owner_arrays, script_post, script_n):
        from wwhip import _lib
        self.lib = _lib.load()
        self.post, self.n = script_post, script_n  # [T][S][2], [T][S]
        self.t = 0
        self.resets = []
        self.arr = owner_arrays  # () -> (is_speech, is_active, post, n_post) arrays the addresses point at

    def step_trigger(self, frames, p_speech, p_active, threshold, st):
        speech, active, post, n = self.arr()
        n[:] = np.where((speech != 0) & (active == 0), self.n[self.t], 0)
        post[:] = np.where(np.arange(2)[None, :] < n[:, None], self.post[self.t], 0.0)
        self.t += 1
        S = len(n)
        assert self.lib.ww_trigger_bank_step(S, p_speech, p_active, st[2], st[3], threshold, st[0], st[1], st[4], st[5], st[6], st[7]) == 0

    def reset(self, ids=None):
        self.resets.append(None if ids is None else list(ids))

    def close(self):
        pass


class _RefTrigger:
    """``WakewordTrigger.__call__`` / ``_detect`` of the reference with the models replaced by scripted posteriors
    (spokestack/wakeword/tflite.py:123-146,228-239), one stream."""

    def __init__(self, threshold):
        self.thr, self.was, self.pmax, self.resets = threshold, False, np.float32(0.0), 0

    def __call__(self, ctx, posts):
        fall = self.was and not ctx.is_speech
        self.was = ctx.is_speech
        if not ctx.is_active and ctx.is_speech:
            for p in posts:
                if p > self.pmax:
                    self.pmax = p
                if float(p) > self.thr and not ctx.is_active:
                    ctx.is_active = True
        if fall:
            self.pmax = np.float32(0.0)
            self.resets += 1


def _bank_script(S, T, seed):
    rng = np.random.default_rng(seed)
    raw = np.zeros((T, S), bool)
    for s in range(S):
        t, v = 0, bool(rng.random() < 0.5)
        while t < T:
            n = int(rng.integers(1, 30))
            raw[t:t + n, s] = v
            t, v = t + n, not v
    post = rng.random((T, S, 2)).astype(np.float32) ** 4  # mostly small, now and then above the threshold
    n = rng.integers(0, 3, (T, S)).astype(np.int32)
    return raw, post, n


def test_banked_stages_equal_the_single_stream_stages():
    """VadBank -> WakewordBank -> ActivationTimeoutBank on a ContextBank (one library pass per stage and tick) against S chains of
    the single-stream classes on SpeechContext objects: the same flags after every stage of every tick, the same activate /
    deactivate events in the same order, the same running maxima and resets.  Host logic only (scripted posteriors)."""
    from wwhip.activation_timeout import ActivationTimeoutBank
    from wwhip.context import ContextBank
    from wwhip.vad import VadBank, VoiceActivityDetector
    from wwhip.wakeword import WakewordBank
    S, T, THR = 37, 400, 0.6
    raw, post, n = _bank_script(S, T, 5)
    kw_vad = dict(frame_width=20, vad_rise_delay=40, vad_fall_delay=60)
    kw_to = dict(frame_width=20, min_active=100, max_active=400)
    cb = ContextBank(S)
    logs = [[] for _ in range(S)]
    for s in range(0, S, 2):  # every other stream has handlers (a view exists only where one was asked for)
        for name in ("activate", "deactivate"):
            cb[s].add_handler(name, (lambda s, nm: (lambda c: logs[s].append(nm)))(s, name))
    bank_log = []
    cb.add_handler("activate", lambda c: bank_log.append(("activate", c._s)))
    cb.add_handler("deactivate", lambda c: bank_log.append(("deactivate", c._s)))
    vad, to = VadBank(S, **kw_vad), ActivationTimeoutBank(S, **kw_to)
    wake = WakewordBank(S, posterior_threshold=THR, bank=_StubStreamBank(lambda: (cb.is_speech, cb.is_active, wake._post, wake._n), post, n))
    woke = []
    wake._on_wake = lambda ids: woke.extend(int(i) for i in ids)

    ref_ctx = [SpeechContext() for _ in range(S)]
    ref_logs = [[] for _ in range(S)]
    ref_all = []
    for s in range(S):
        for name in ("activate", "deactivate"):
            ref_ctx[s].add_handler(name, (lambda s, nm: (lambda c: (ref_logs[s].append(nm), ref_all.append((nm, s)))))(s, name))
    tick = {"t": 0}
    ref_vad = [VoiceActivityDetector(classifier=(lambda s: (lambda fb, sr: bool(raw[tick["t"], s])))(s), **kw_vad) for s in range(S)]
    ref_trig = [_RefTrigger(THR) for _ in range(S)]
    ref_to = [ActivationTimeout(**kw_to) for _ in range(S)]
    frame = np.zeros(320, np.int16)
    frames = np.zeros((S, 320), np.int16)
    for t in range(T):
        tick["t"] = t
        n_ref = len(ref_all)
        # stage by stage, all streams (events of one stage come out in stream order, as the bank raises them)
        vad(cb, frames, raw=raw[t])
        for s in range(S):
            ref_vad[s](ref_ctx[s], frame)
        assert [bool(c.is_speech) for c in ref_ctx] == list(cb.is_speech.astype(bool)), t
        got_post = wake.step(cb, frames)
        for s in range(S):
            sampled = ref_ctx[s].is_speech and not ref_ctx[s].is_active
            ref_trig[s](ref_ctx[s], post[t, s, :n[t, s]])
            assert wake.n_post[s] == (n[t, s] if sampled else 0)
            assert np.array_equal(got_post[s, :wake.n_post[s]], post[t, s, :wake.n_post[s]])
        assert [bool(c.is_active) for c in ref_ctx] == list(cb.is_active.astype(bool)), t
        assert np.array_equal(wake.posterior_max, np.array([r.pmax for r in ref_trig], np.float32)), t
        to(cb, frames)
        for s in range(S):
            ref_to[s](ref_ctx[s], frame)
        assert [bool(c.is_active) for c in ref_ctx] == list(cb.is_active.astype(bool)), t
        assert bank_log[n_ref:] == ref_all[n_ref:], t
    for s in range(S):
        assert logs[s] == (ref_logs[s] if s % 2 == 0 else [])
        assert cb[s].is_active == ref_ctx[s].is_active and cb[s].is_speech == ref_ctx[s].is_speech
    assert woke == [s for (nm, s) in ref_all if nm == "activate"]
    n_resets = sum(len(r) for r in wake._bank.resets)
    assert n_resets == 0  # (the stub is not asked: ww_stream_step_trigger resets inside the library; the ids are in wake._fall)
    acts = sum(nm == "activate" for nm, _ in ref_all)
    assert acts >= S and sum(nm == "deactivate" for nm, _ in ref_all) >= acts - S  # the script exercises both edges, many times


def test_activation_timeout_bank_replays_reference_trace(golden):
    """The recorded trace of the reference's ActivationTimeout (tests/golden/activation_timeout_trace.json) through the banked
    stage: every case as one stream of ONE bank per parameter set would need equal lengths, so each case runs as a bank of three
    identical streams - same is_active after every frame, same events, on all three."""
    from wwhip.activation_timeout import ActivationTimeoutBank
    from wwhip.context import ContextBank
    cases = json.load(open(os.path.join(golden, "activation_timeout_trace.json")))
    for case in cases:
        cb = ContextBank(3)
        events = [[] for _ in range(3)]
        for s in range(3):
            for name in ("activate", "deactivate"):
                cb[s].add_handler(name, (lambda s, n: (lambda c: events[s].append(n)))(s, name))
        stage = ActivationTimeoutBank(3, **case["kwargs"])
        got = [[] for _ in range(3)]
        for t, (sp, act, rst, want) in enumerate(zip(case["is_speech"], case["activate"], case["reset"], case["is_active"])):
            cb.is_speech[:] = sp == "1"
            if act == "1":
                for c in cb:
                    c.is_active = True
            if rst == "1":
                stage.reset()
            n_ev = [len(e) for e in events]
            stage(cb, None)
            assert list(cb.is_active) == [int(want == "1")] * 3, (case["kwargs"], t)
            for s in range(3):
                got[s] += [[t, e] for e in events[s][n_ev[s]:]]
        assert got == [case["events"]] * 3


def test_a_quiet_tick_costs_no_per_stream_python():
    """A tick in which no stream changes state makes no per-stream Python call: the number of Python-level function calls of
    VadBank -> WakewordBank -> ActivationTimeoutBank on a ContextBank is the same for 8 streams and for 512, and small; a tick
    in which k streams change adds calls for those k only."""
    import sys
    from wwhip.activation_timeout import ActivationTimeoutBank
    from wwhip.context import ContextBank
    from wwhip.vad import VadBank
    from wwhip.wakeword import WakewordBank

    def calls_per_tick(S, loud_at=None):
        T = 12
        post = np.full((T, S, 2), 0.1, np.float32)
        n = np.full((T, S), 2, np.int32)
        if loud_at is not None:
            post[loud_at, :3, 1] = 0.9  # three streams fire in that tick
        cb = ContextBank(S)
        handled = []
        for s in range(S):
            cb[s].add_handler("activate", lambda c: handled.append(c._s))
        wake = WakewordBank(S, posterior_threshold=0.5, bank=_StubStreamBank(lambda: (cb.is_speech, cb.is_active, wake._post, wake._n), post, n))
        vad, to = VadBank(S), ActivationTimeoutBank(S)
        frames, raw = np.zeros((S, 320), np.int16), np.ones(S, bool)
        counts = []
        for t in range(T):
            k = [0]

            def prof(frame, event, arg):
                if event == "call":  # Python-level calls (the stub bank's two included: constant in S)
                    k[0] += 1
            sys.setprofile(prof)
            try:
                vad(cb, frames, raw=raw)
                wake.step(cb, frames)
                to(cb, frames)
            finally:
                sys.setprofile(None)
            counts.append(k[0])
        return counts, handled

    small, _ = calls_per_tick(8)
    large, _ = calls_per_tick(512)
    assert small[2:] == large[2:], (small, large)       # (the first ticks bind the stages to the ContextBank)
    assert max(large[2:]) <= 24, large                   # three stage calls and their few helpers (13 today), whatever S is
    loud, handled = calls_per_tick(512, loud_at=6)
    assert handled == [0, 1, 2]
    quiet = large[5]
    assert loud[5] == quiet and loud[7] == quiet and quiet < loud[6] <= quiet + 3 * 8, loud


def test_chunk_staging_through_the_extension_equals_the_per_file_path():
    """wwhip.evaluate._prep_chunk on a list of clips in memory: the per-clip bookkeeping in ONE call of the CPython extension
    (csrc/hostext.c, scan_pcm16: address, length, "contiguous int16" by the buffer protocol) gives the same copy runs and offset
    tables as the per-file path it replaces; a list that holds anything else (a float clip, a strided view) falls back to that
    path by itself; a clip shorter than the plan says is refused before any address leaves for the copy threads (the advisor's
    out-of-bounds read of round 4); and the staged bytes are what the runs say (ww_host_stage_i16, no GPU)."""
    from wwhip import _lib, _wwhostext
    from wwhip import evaluate as E

    class Ctx:
        device = 0

    class Eng:
        window, ctx = 151, Ctx()

    rng = np.random.default_rng(8)
    clips = [rng.integers(-3000, 3000, int(n), dtype=np.int16) for n in rng.integers(9000, 30000, 60)]

    def chunks_of(files, lengths=None, generic=False, world=3, rank=1):
        ph = E._Phases(None)
        job = E._PosteriorJob(Eng, "false_negatives", files, 20, 16000, rank, world, None, lengths, True, ph, None, first_chunk=1 << 17)
        if generic:
            job.raw_list = None
        for ch in job.chunks:
            E._prep_chunk(ch, ph)
        return job

    fast, slow = chunks_of(E._Int16Clips(clips)), chunks_of(E._Int16Clips(clips), generic=True)
    assert fast.raw_list is not None and len(fast.chunks) == len(slow.chunks) >= 3
    for a, b in zip(fast.chunks, slow.chunks):
        for x, y in zip(a.copy, b.copy):
            np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(a.soffs, b.soffs)
        np.testing.assert_array_equal(a.foffs, b.foffs)
        assert (a.nf_max, a.total_f) == (b.nf_max, b.total_f) and a.host_pieces is None and b.host_pieces is None
    # the staged samples of a chunk: piece p of file k = the padded stream's samples [s0, s1) (0.5 s of zeros around every clip)
    ch = fast.chunks[1]
    d, pp, c = ch.copy
    total = int(ch.soffs[-1])
    dst = np.full(total, 77, np.int16)
    assert _lib.load().ww_host_stage_i16(_lib.ptr(dst), total, len(d), _lib.ptr(d), _lib.ptr(pp), _lib.ptr(c), 0, total, 2) == 0
    plan = fast.plan
    whole = np.zeros(int(plan.padded.sum()), np.int16)
    for k, x in enumerate(clips):
        whole[plan.pos[k] + plan.pad: plan.pos[k] + plan.pad + len(x)] = x
    for p, (k, i0, i1) in enumerate(np.asarray(ch.runs).tolist()):
        s0, s1 = plan.sample_range(k, i0, i1)
        np.testing.assert_array_equal(dst[ch.soffs[p]:ch.soffs[p + 1]], whole[s0:s1])
    # a list with other things in it: the extension takes what it can, the job goes back to the per-file path
    mixed = list(clips)
    mixed[7] = clips[7].astype(np.float32) / np.float32(32768.0)
    m = chunks_of(mixed, world=1, rank=0)
    assert m.raw_list is None and any(ch.host_pieces is not None for ch in m.chunks)  # (float clips: staged through NumPy, as before)
    strided = list(clips)
    strided[3] = np.repeat(clips[3], 2)[::2]
    s = chunks_of(strided, world=1, rank=0)
    assert s.raw_list is None
    for a, b in zip(s.chunks, chunks_of(E._Int16Clips(clips), generic=True, world=1, rank=0).chunks):
        np.testing.assert_array_equal(a.copy[0], b.copy[0])
        np.testing.assert_array_equal(a.copy[2], b.copy[2])
    # a declared length larger than the array (a truncated wav whose header says more): refused, in both paths
    lengths = [len(x) for x in clips]
    lengths[20] += 5000
    for generic in (False, True):
        with pytest.raises(ValueError, match="file 20 holds"):
            chunks_of(E._Int16Clips(clips), lengths=lengths, generic=generic, world=1, rank=0)
    # the extension's own contract
    import sys
    x5 = clips[5]
    before = sys.getrefcount(x5)
    picked, lens = _wwhostext.take(clips, np.array([5, 0, 59, 5], np.int64))
    held = sys.getrefcount(x5)
    same = all(p is clips[i] for p, i in zip(picked, (5, 0, 59, 5)))
    del picked
    after = sys.getrefcount(x5)
    assert same and (held, after) == (before + 2, before)      # the list holds its own references and gives them back
    assert np.frombuffer(lens, np.int64).tolist() == [len(clips[5]), len(clips[0]), len(clips[59]), len(clips[5])]
    assert _wwhostext.take((), np.zeros(0, np.int64))[0] == []
    with pytest.raises(IndexError):
        _wwhostext.take(clips, np.array([60], np.int64))
    with pytest.raises(TypeError):
        _wwhostext.take([1, 2], np.array([0], np.int64))      # an item without len(): its own error, nothing leaked
    with pytest.raises(TypeError):
        _wwhostext.take(clips, np.array([0], np.int32))        # idx must be int64
    addr, ns, seen = np.zeros(4, np.int64), np.zeros(4, np.int64), np.zeros(4, np.uint8)
    things = [clips[0], clips[1][::2], [1, 2, 3], np.zeros((4, 4), np.int16)]
    assert _wwhostext.scan_pcm16(things, np.arange(4), addr, ns, seen) == 1
    assert seen.tolist() == [1, 0, 0, 0] and addr[0] == clips[0].ctypes.data and ns[0] == len(clips[0])
    with pytest.raises(IndexError):
        _wwhostext.scan_pcm16(things, np.array([4], np.int64), addr, ns, seen)
    with pytest.raises(ValueError):
        _wwhostext.scan_pcm16(things, np.arange(4), addr[:3], ns, seen)


@pytest.mark.timeout(300)
@pytest.mark.parametrize("sanitizer", ["thread", "address,undefined"])
def test_host_staging_code_under_sanitizers(tmp_path, sanitizer):
    """csrc/host_stage.h - the uploader's copy-thread pool (threads woken per chunk, a bounded spin in front of every sleep: round 6),
    the streaming-store copy loops and the run bookkeeping - compiled alone with ThreadSanitizer and with Address + UB sanitizer
    and hammered on the CPU (tests/native/host_stage_check.cpp: random chunks staged in slices by the pool against a plain loop,
    hand-offs with and without pauses so that spinning AND sleeping waiters are exercised).  The GPU boxes run no sanitizers;
    this is where the library's only lock-free hand-off is checked."""
    import shutil
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cxx = "/opt/rocm/lib/llvm/bin/clang++"  # (the staging loops use clang's __builtin_nontemporal_store, as hipcc compiles them)
    if not os.path.exists(cxx):
        cxx = shutil.which("clang++")
    if cxx is None:
        pytest.skip("no clang++ in this image")
    exe = tmp_path / "host_stage_check"
    b = subprocess.run([cxx, "-std=c++17", "-O1", "-g", f"-fsanitize={sanitizer}", "-pthread", "-I" + os.path.join(root, "wakeword-detection_amd", "csrc"),
                        os.path.join(root, "tests", "native", "host_stage_check.cpp"), "-o", str(exe)], capture_output=True, text=True)
    if b.returncode != 0 and "sanitizer" in (b.stderr + b.stdout).lower():
        pytest.skip("this clang has no sanitizer runtime: " + b.stderr[-300:])
    assert b.returncode == 0, b.stderr[-2000:]
    r = subprocess.run([str(exe), "150"], capture_output=True, text=True, timeout=240,
                       env=dict(os.environ, TSAN_OPTIONS="halt_on_error=1 exitcode=66", ASAN_OPTIONS="detect_leaks=1", UBSAN_OPTIONS="halt_on_error=1"))
    assert r.returncode == 0 and r.stdout.startswith("ok "), (r.returncode, r.stdout[-500:], r.stderr[-3000:])
    assert "Sanitizer" not in r.stderr, r.stderr[-3000:]


def test_pipeline_bank_and_context_bank_surface():
    """SpeechPipelineBank keeps SpeechPipeline's control surface (spokestack/pipeline.py:30-111) over a ContextBank: start / stop /
    pause / resume / step / run / cleanup / event; stages are called once per tick with (contexts, frames) in list order; the
    per-tick "step" event reaches the bank-wide handler; activate() / deactivate() raise each stream's event once; ContextBank
    views behave like SpeechContext objects (reset, transcript, confidence) and index like a sequence."""
    from wwhip.context import ContextBank, SpeechContext
    from wwhip.pipeline import SpeechPipelineBank
    S = 5
    log = []

    class Source:
        def __init__(self):
            self.t = 0

        def read(self):
            self.t += 1
            return np.full((S, 320), self.t, np.int16)

        def start(self):
            log.append("src.start")

        def stop(self):
            log.append("src.stop")

        def close(self):
            log.append("src.close")

    class Stage:
        def __init__(self, name):
            self.name = name

        def __call__(self, contexts, frames):
            log.append((self.name, len(contexts), int(frames[0, 0])))
            if self.name == "b" and frames[0, 0] == 2:
                pipe.stop()  # a stage may stop the loop (run() returns after this tick)

        def close(self):
            log.append(self.name + ".close")

    pipe = SpeechPipelineBank(Source(), [Stage("a"), Stage("b")], S)
    ctx = pipe.context
    assert isinstance(ctx, ContextBank) and len(ctx) == S and isinstance(ctx[2], SpeechContext) and ctx[2] is ctx[2]
    assert [c._s for c in ctx] == list(range(S))
    with pytest.raises(IndexError):
        ctx[S]
    events = []
    pipe.event(lambda bank: events.append(("step", bank.S)), name="step")

    @pipe.event
    def on_activate(c):
        events.append(("activate", c._s))

    pipe.event(lambda c: events.append(("deactivate", c._s)), name="deactivate")
    ctx[3].add_handler("activate", lambda c: events.append(("own", c._s)))
    pipe.start(); pipe.start()
    assert pipe.is_running and log == ["src.start"]
    pipe.pause(); pipe.step(); pipe.resume()
    assert log == ["src.start", "src.stop", "src.start"] and events == [("step", S)]   # a paused step reads nothing, runs no stage
    pipe.run()                                                                          # two ticks, then stage b stops the loop
    assert [e for e in log if isinstance(e, tuple)] == [("a", S, 1), ("b", S, 1), ("a", S, 2), ("b", S, 2)]
    assert log[-4:] == ["src.stop", "a.close", "b.close", "src.close"] and not pipe.is_running
    assert events.count(("step", S)) == 3
    # activation from outside: one event per stream, the stream's own handler first
    del events[:]
    ctx[3].is_active = True
    assert events == [("own", 3), ("activate", 3)] and ctx.is_active.tolist() == [0, 0, 0, 1, 0]
    pipe2 = SpeechPipelineBank(Source(), [], 3)
    seen = []
    pipe2.event(lambda c: seen.append(c._s), name="activate")
    pipe2.activate(); pipe2.activate()
    assert seen == [0, 1, 2] and pipe2.context.is_active.all()
    pipe2.event(lambda c: seen.append(-c._s - 1), name="deactivate")
    pipe2.deactivate()
    assert seen[3:] == [-1, -2, -3]
    # views are SpeechContexts: state of their own, reset through the bank
    v = ctx[1]
    v.transcript, v.confidence, v.is_speech = "hi", 0.5, True
    assert ctx.is_speech.tolist() == [0, 1, 0, 0, 0]
    del events[:]
    ctx.reset()
    assert not ctx.is_speech.any() and not ctx.is_active.any() and (v.transcript, v.confidence) == ("", 0.0)
    assert events == [("deactivate", 3)]                      # only the stream that WAS active hears about it
