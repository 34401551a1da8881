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


def test_library_carries_gfx950_code(tmp_path):
    import shutil
    from wwhip import _lib
    # llvm-objdump --offloading unbundles the code objects NEXT TO its input: work on a copy in a scratch directory
    lib = shutil.copy(_lib.LIB_PATH, tmp_path / "libwwhip.so")
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "--offloading", str(lib)], capture_output=True, text=True,
                         cwd=tmp_path)
    if out.returncode == 0 and out.stdout:
        assert "gfx950" in out.stdout


def test_no_gpu_means_loud_failure():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from wwhip import _lib
    with pytest.raises(RuntimeError):
        _lib.Context(0)


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
