"""C-ABI library: loads, exports every symbol include/wwhip.h declares, fails loudly without
a GPU (no compute calls here).  Plus the world_size-2 gloo test of the sharding/gather path."""
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def header_symbols():
    text = open(os.path.join(ROOT, "include", "wwhip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(ww_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from wwhip import _lib
    lib = _lib.load()
    declared = header_symbols()
    assert len(declared) >= 28
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/wwhip.h but not exported"
    assert sorted(_lib.SYMBOLS) == declared
    assert b"gfx950" in lib.ww_version()
    assert lib.ww_num_frames(24000, 160) == 147 and lib.ww_num_frames(511, 160) == 0
    from wwhip import _wwhostext  # the CPython extension built next to it (csrc/hostext.c): host bookkeeping, no GPU code
    assert callable(_wwhostext.scan_pcm16)


def test_no_exception_can_leave_the_c_abi():
    """"all functions return int status 0/negative, never throw".  Every exported definition under csrc/ either
    has its whole body between WW_GUARD_BEGIN and WW_GUARD_END (std::bad_alloc -> WW_ENOMEM, anything else -> WW_EINTERNAL,
    csrc/common.h) or is marked WW_NOTHROW (a constant or a field read); the create functions hold what they allocated in a
    ww_scoped owner, so that an early exit frees it."""
    csrc = os.path.join(ROOT, "wakeword-detection_amd", "csrc")
    text = {f: open(os.path.join(csrc, f)).read() for f in sorted(os.listdir(csrc)) if f.endswith(".hip")}
    seen = {}
    for name in header_symbols():
        hits = [(f, m) for f, t in text.items() for m in re.finditer(r"^[A-Za-z_][A-Za-z0-9_ \*]*\b" + name + r"\(", t, flags=re.M)]
        assert len(hits) == 1, (name, [f for f, _ in hits])
        f, m = hits[0]
        t = text[f]
        open_brace = t.index("{", m.end())  # (no parameter list under csrc/ holds a brace)
        head = t[m.start():open_brace]
        eol = t.index("\n", open_brace)
        end = eol - 1 if t[eol - 1] == "}" else t.index("\n}\n", open_brace)  # a one-line definition, or the brace in column 0
        body = t[open_brace + 1:end].strip()
        if "WW_NOTHROW" in head:
            assert "new " not in body and "push_back" not in body and "std::" not in body, name
            seen[name] = "nothrow"
            continue
        assert body.startswith("WW_GUARD_BEGIN"), f"{name} ({f}): body does not open with WW_GUARD_BEGIN"
        assert re.search(r"WW_GUARD_END\([^\n]*\)$", body), f"{name} ({f}): body does not close with WW_GUARD_END(ctx)"
        assert body.count("WW_GUARD_BEGIN") == 1 and body.count("WW_GUARD_END(") == 1, name
        seen[name] = "guard"
    assert sum(v == "nothrow" for v in seen.values()) == 4, seen  # ww_version, ww_last_error, ww_ctx_stream, ww_num_frames
    for f, name in (("api.hip", "ww_ctx"), ("api.hip", "ww_model"), ("streams.hip", "ww_streams"), ("uploader.hip", "ww_uploader")):
        assert re.search(r"ww_scoped<" + name + r", ", text[f]), f"{f}: the {name} create function does not own its object"
    from wwhip import _lib
    assert _lib.WW_EINTERNAL == -7


def test_library_carries_gfx950_code(tmp_path):
    import shutil
    from wwhip import _lib
    # llvm-objdump --offloading unbundles the code objects NEXT TO its input: work on a copy in a scratch directory
    lib = shutil.copy(_lib.LIB_PATH, tmp_path / "libwwhip.so")
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", str(lib)], capture_output=True, text=True,
                         cwd=tmp_path)
    if out.returncode == 0 and out.stdout:
        assert "gfx950" in out.stdout


def test_no_kernel_spills_or_keeps_arrays_in_scratch_memory(tmp_path):
    """The code objects inside libwwhip.so, read back with llvm-readelf: every kernel's private segment is empty and no
    register was spilled.  The kernels are written around their register budgets; a local array the compiler
    leaves in scratch memory (arrays of HIP's struct vector types do, clang ext-vectors do not; so does a constant it takes
    the address of) turns a prologue into global-memory round trips without any warning - round 4 lost 10 % of the
    split-bf16 Wavenet that way before the metadata was looked at."""
    import shutil
    from wwhip import _lib
    readelf, objdump = "/opt/rocm/lib/llvm/bin/llvm-readelf", "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not (os.path.exists(readelf) and os.path.exists(objdump)):
        pytest.skip("no llvm-readelf / llvm-objdump in this image")
    lib = shutil.copy(_lib.LIB_PATH, tmp_path / "libwwhip.so")
    subprocess.run([objdump, "--offloading", str(lib)], capture_output=True, text=True, cwd=tmp_path, check=True)
    objs = sorted(f for f in os.listdir(tmp_path) if "amdgcn" in f and "gfx950" in f)
    assert len(objs) >= 5, objs                         # one per source file with kernels
    kernels, bad = 0, []
    for f in objs:
        notes = subprocess.run([readelf, "--notes", str(tmp_path / f)], capture_output=True, text=True, check=True).stdout
        name = None
        for line in notes.splitlines():
            m = re.match(r"\s*\.(name|private_segment_fixed_size|vgpr_spill_count|sgpr_spill_count):\s*(\S+)", line)
            if not m:
                continue
            if m.group(1) == "name":
                name, kernels = m.group(2), kernels + 1
            elif int(m.group(2)) != 0:
                bad.append((name, m.group(1), int(m.group(2))))
    assert kernels >= 40 and not bad, bad


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from wwhip import _lib
    with pytest.raises(RuntimeError):
        _lib.Context(0)


def test_a_device_index_that_does_not_exist_is_an_error_message_not_an_abort():
    """The first lines of the multi-device path that can run anywhere: a context on a device index at or past the visible
    count (a rank whose LOCAL_RANK exceeds the node's GPUs) fails with the library's own text - "no HIP device visible" in a
    container without a GPU, "device N out of range" on a GPU box - never with a HIP abort; negative indices likewise."""
    import torch
    from wwhip import _lib
    n = torch.cuda.device_count() if torch.cuda.is_available() else 0
    for dev in (n, n + 7, -1):
        if n == 0:
            with pytest.raises(RuntimeError, match="no HIP device visible"):
                _lib.Context(dev)
        else:
            with pytest.raises(ValueError, match=r"device -?\d+ out of range \(0\.\.%d\)" % (n - 1)):
                _lib.Context(dev)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "wakeword-detection_amd")
    for d, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                text = open(os.path.join(d, f), errors="replace").read()
                assert "import oracle" not in text and "from oracle" not in text and "ww_oracle" not in text, f


def test_shard_helpers():
    from wwhip import dist as D
    lens = [5, 100, 7, 50, 60, 1]
    sh = D.shard_by_length(lens, 2)
    assert sorted(sh[0] + sh[1]) == list(range(6))
    assert sh[0][0] == 1 and sh[1][0] == 4
    assert D.split_stream(10, 3) == [(0, 4), (4, 7), (7, 10)]


_WORKER = r"""
import os, sys
sys.path[:0] = [{root!r}, os.path.join({root!r}, "wakeword-detection_amd")]
import numpy as np, torch, torch.distributed as dist
from wwhip import dist as D
rank = int(os.environ["RANK"]); world = int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
lens = [300, 20, 150, 90, 10, 400, 33]
mine = D.shard_by_length(lens, world)[rank]
local = np.array([lens[i] * 0.001 + i for i in mine], np.float32)      # stand-in posteriors
full = D.gather_posteriors(local, mine, len(lens))
want = np.array([l * 0.001 + i for i, l in enumerate(lens)], np.float32)
assert np.allclose(full, want), (full, want)
lo, hi = D.split_stream(11, world)[rank]
full2 = D.gather_posteriors(np.arange(lo, hi, dtype=np.float32), list(range(lo, hi)), 11)
assert np.array_equal(full2, np.arange(11, dtype=np.float32))
dist.barrier(); dist.destroy_process_group()
open(os.path.join({out!r}, f"rank{{rank}}.ok"), "w").write("ok")
"""


def test_posterior_gather_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER.format(root=ROOT, out=str(tmp_path)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29541")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                        "--master-addr", "127.0.0.1", "--master-port", "29541", str(script)],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert (tmp_path / "rank0.ok").exists() and (tmp_path / "rank1.ok").exists()


_PLAN_WORKER = """
import os, sys
sys.path[:0] = [{root!r}, os.path.join({root!r}, "wakeword-detection_amd")]
import numpy as np, torch.distributed as dist
from wwhip import dist as D
from wwhip.evaluate import StreamPlan
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
plan = StreamPlan([30000, 41000, 4000, 24000, 51234], 151)
for eval_type in ("false_negatives", "false_accepts"):
    slots, vals = [], []
    for k, i0, i1 in plan.shares(eval_type, world)[rank]:
        s0, s1 = plan.sample_range(k, i0, i1)
        assert (s1 - s0 - 512) // 160 + 1 == 2 * (i1 - i0 - 1) + 151      # the frames this rank's piece yields = what its windows need
        slots.append(np.arange(plan.offs[k] + i0, plan.offs[k] + i1))
        vals.append((plan.F[k] + 2 * np.arange(i0, i1)) * 0.25)           # a "posterior" that names its first global frame
    slots = np.concatenate(slots) if slots else np.zeros(0, np.int64)
    vals = np.concatenate(vals) if vals else np.zeros(0)
    full = D.gather_posteriors(vals.astype(np.float32), slots, plan.total)
    want = np.concatenate([(plan.F[k] + 2 * np.arange(plan.n_win[k])) * 0.25 for k in range(5)]).astype(np.float32)
    assert np.array_equal(full, want), eval_type
    # the product's own exchange (wwhip.evaluate._PosteriorJob.finish: values only, every rank derives the slots from the plan)
    from wwhip import evaluate as E
    class Eng: window = 151
    lens = [30000, 41000, 14000, 24000, 51234, 9000, 33000]
    job = E._PosteriorJob(Eng, eval_type, [np.zeros(n, np.int16) for n in lens], 20, 16000, rank, world, None, None, True,
                          E._Phases(None), None)
    p2 = job.plan
    assert (p2.n_win > 0).all()
    assert [tuple(r) for r in job.mine] == [tuple(r) for r in p2.shares(eval_type, world)[rank]]
    assert [tuple(r) for ch in job.chunks for r in ch.runs.tolist()] == [tuple(r) for r in job.mine.tolist()]  # the share, cut into chunks, nothing lost
    name = lambda k, i0, i1: (p2.F[k] + 2 * np.arange(i0, i1)) * 0.25           # a "posterior" that names its first global frame
    # what a rank contributes (the device picks it: ww_posterior_pick_dev): every window's value for the negative stream, the
    # maximum of every run for the wake-word clips
    mine = [name(k, i0, i1) for k, i0, i1 in job.mine]
    if eval_type == "false_negatives":
        mine = [np.array([v.max()]) for v in mine]
    job.vals = np.concatenate(mine).astype(np.float32) if len(mine) else np.zeros(0, np.float32)
    assert len(job.vals) == job.n_values(job.mine)
    got = job.finish(None, E._Phases(None), True)
    every = [name(k, 0, p2.n_win[k]).astype(np.float32) for k in range(len(lens))]
    want2 = np.array([v.max() for v in every], np.float32) if eval_type == "false_negatives" else np.concatenate(every)
    assert np.array_equal(got, want2), (eval_type, got[:8], want2[:8])
dist.barrier(); dist.destroy_process_group()
open(os.path.join({out!r}, f"plan{{rank}}.ok"), "w").write("ok")
"""


def test_stream_plan_shares_and_gather_gloo_world3(tmp_path):
    """SURVEY 8(e) on CPU: the window shares of the sharded reference flow (whole files for the positives, contiguous
    posterior ranges of the negative stream) + the posterior gather reassemble the full list on every rank (3 ranks)."""
    script = tmp_path / "plan_worker.py"
    script.write_text(_PLAN_WORKER.format(root=ROOT, out=str(tmp_path)))
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3",
                        "--master-addr", "127.0.0.1", "--master-port", "29543", str(script)],
                       capture_output=True, text=True, env=dict(os.environ), timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert all((tmp_path / f"plan{k}.ok").exists() for k in range(3))


def test_bench_parent_starts_ranks_without_touching_a_gpu():
    """`bench.py --gpus 2` with no WORLD_SIZE spawns its ranks itself.  Without a GPU the children refuse to run
    ("no CPU fallback"), and the parent must report that as a failure instead of printing a line."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_bench_eval.py")
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode != 0 and r.stdout.strip() == ""
    assert "rank exit codes" in r.stderr and "needs an MI355X" in r.stderr


@pytest.mark.timeout(120)
def test_default_context_fails_fast_without_a_gpu(tmp_path):
    """`_lib.default_context` (what every drop-in class goes through) must raise - not hang - when no device is usable; in a
    child process with a deadline, because a lock-order mistake here would otherwise stall the whole suite."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: the context is created for real by the -m gpu tests")
    code = ("import sys; sys.path[:0] = [%r, %r]\n"
            "from wwhip import _lib\n"
            "try:\n    _lib.default_context(0)\nexcept RuntimeError as e:\n    print('raised', e)\n" % (ROOT, os.path.join(ROOT, "wakeword-detection_amd")))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=90)
    assert r.returncode == 0 and "raised" in r.stdout, r.stdout + r.stderr


def test_a_file_shorter_than_its_planned_length_is_refused():
    """The copy runs of a chunk take their counts from the plan and hand raw addresses to the library's copy threads: a
    plan built on more samples than a file holds (a truncated wav whose header promises more, a caller's own `lengths`) must
    be refused on the host - StreamPlan([30000]) over a 20,000-sample array used to mean a 20,000-byte read past its end."""
    import types
    from wwhip import evaluate as E
    plan = E.StreamPlan([30000], 151)
    ch = E._Chunk()
    ch.runs = plan.shares("false_negatives", 1)[0]
    ch.n_win = np.array([i1 - i0 for _, i0, i1 in ch.runs], np.int64)
    ch.host_pieces = None
    for dtype in (np.int16, np.float32):
        ch.job = types.SimpleNamespace(plan=plan, load=lambda f: np.zeros(20000, dtype))
        with pytest.raises(ValueError, match="20000 samples"):
            E._prep_chunk(ch, E._Phases(None))
    ch.job = types.SimpleNamespace(plan=plan, load=lambda f: np.zeros(30000, np.int16))
    E._prep_chunk(ch, E._Phases(None))
    assert int(ch.copy[2].sum()) == 30000


def test_host_staging_equals_the_padded_stream():
    """ww_host_stage_i16 + wwhip.evaluate._piece_runs (what a rank uploads: the samples its windows are functions of, piece
    after piece, written once by the library's host threads) against the literal construction - the padded files laid end to
    end (never-reset ring) or each on its own - for positives and negative streams, several world sizes, a joined stream
    (JoinedPCM: one run per clip) and the multi-threaded path; overlapping runs are refused.  Host code only: no GPU call."""
    import ctypes as C
    from wwhip import _lib
    from wwhip import evaluate as E
    lib = _lib.load()
    rng = np.random.default_rng(0)

    def stage(plan, runs, data, threads):
        soffs, d, pp, c = E._piece_runs(plan, runs, data)
        need = int(soffs[-1]) + 16
        buf = np.full(need, -7, np.int16)
        assert lib.ww_host_stage_i16(buf.ctypes.data, need, len(d), _lib.ptr(d), _lib.ptr(pp), _lib.ptr(c), 0, need, threads) == 0
        return buf

    for trial in range(12):
        clips = [rng.integers(-3000, 3000, int(n)).astype(np.int16) for n in rng.integers(100, 60000, int(rng.integers(1, 9)))]
        for carry in (True, False):
            plan = E.StreamPlan([len(c) for c in clips], 151, 320, 16000, 2, carry)
            whole = np.zeros(int(plan.padded.sum()), np.int16)
            for k, x in enumerate(clips):
                whole[plan.pos[k] + 8000: plan.pos[k] + 8000 + len(x)] = x
            for et, world in (("false_negatives", 1), ("false_negatives", 3), ("false_accepts", 2)):
                for rank in range(world):
                    runs = plan.shares(et, world)[rank]
                    if not runs:
                        continue
                    data = {f: clips[f] for k, _, _ in runs for f in ((k - 1, k) if carry and k > 0 else (k,))}
                    want = []
                    for k, i0, i1 in runs:
                        s0, s1 = plan.sample_range(k, i0, i1)
                        if carry:
                            want.append(whole[s0:s1])
                        else:
                            one = np.zeros(int(plan.padded[k]), np.int16)
                            one[8000:8000 + len(clips[k])] = clips[k]
                            want.append(one[s0:s1])
                    np.testing.assert_array_equal(stage(plan, runs, data, 3), np.concatenate(want + [np.zeros(16, np.int16)]))
    clips = [rng.integers(-3000, 3000, int(n)).astype(np.int16) for n in rng.integers(100, 60000, 12)]
    j, arr = E.join_negatives_lazy(clips, 9), E.join_negatives(clips, 9)
    plan = E.StreamPlan([len(j)], 151)
    whole = np.zeros(int(plan.padded[0]), np.int16)
    whole[8000:8000 + len(arr)] = arr
    for world in (1, 2, 5):
        for rank in range(world):
            runs = plan.shares("false_accepts", world)[rank]
            s0, s1 = plan.sample_range(*runs[0])
            np.testing.assert_array_equal(stage(plan, runs, {0: j}, 4)[:-16], whole[s0:s1])
    big = rng.integers(-3000, 3000, 5_000_000).astype(np.int16)          # above the one-thread limit: 8 threads, 2 runs
    d, c = np.array([100, 3_000_000], np.int64), np.array([2_000_000, 1_500_000], np.int64)
    pp = np.array([big.ctypes.data, big.ctypes.data + 2 * 2_500_000], np.int64)
    buf = np.full(6_000_000, -7, np.int16)
    for lo, hi in ((0, 1_000_001), (1_000_001, 3_500_000), (3_500_000, 6_000_000)):   # in slices, as the uploader calls it
        assert lib.ww_host_stage_i16(buf.ctypes.data, len(buf), 2, _lib.ptr(d), _lib.ptr(pp), _lib.ptr(c), lo, hi, 8) == 0
    want = np.zeros(6_000_000, np.int16)
    want[100:2_000_100] = big[:2_000_000]
    want[3_000_000:4_500_000] = big[2_500_000:4_000_000]
    np.testing.assert_array_equal(buf, want)
    assert lib.ww_host_stage_i16(buf.ctypes.data, len(buf), 2, _lib.ptr(np.array([100, 50], np.int64)), _lib.ptr(pp), _lib.ptr(c),
                                 0, len(buf), 8) != 0
    assert lib.ww_host_stage_i16(buf.ctypes.data, len(buf), 2, _lib.ptr(d), _lib.ptr(pp), _lib.ptr(c), 10, 5, 8) != 0
