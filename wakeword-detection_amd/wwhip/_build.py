"""Builds ``libwwhip.so`` (gfx950 only) from ``csrc/*.hip`` with hipcc, in-tree.

hipcc cross-compiles without a GPU, so this runs in the build container; the resulting
``.so`` travels with the repository snapshot to the GPU box.
"""
from __future__ import annotations

import os
import shutil
import subprocess
from concurrent.futures import ThreadPoolExecutor
from typing import List

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG_DIR)
CSRC = os.path.join(ROOT, "csrc")
OBJ_DIR = os.path.join(CSRC, "build")
LIB_PATH = os.path.join(PKG_DIR, "libwwhip.so")
HOSTEXT_PATH = os.path.join(PKG_DIR, "_wwhostext.so")  # CPython extension: per-clip bookkeeping of the evaluators' staging (csrc/hostext.c)

SOURCES = ["api.hip", "frontend.hip", "crnn.hip", "wavenet.hip", "posterior.hip", "streams.hip", "uploader.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result"]
FLAGS += os.environ.get("WWHIP_DEFS", "").split()  # development only: e.g. WWHIP_DEFS="-DFPB=32"


def _hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.isfile(cand):
            return cand
    raise RuntimeError("hipcc not found; libwwhip.so cannot be built")


def _stale(target: str, deps: List[str]) -> bool:
    if not os.path.isfile(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(CSRC, "common.h"), os.path.join(CSRC, "fft_device.h"), os.path.join(CSRC, "host_stage.h"),
               os.path.join(os.path.dirname(ROOT), "include", "wwhip.h")]
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ_DIR, s.replace(".hip", ".o"))
        objs.append(obj)
        if force or _stale(obj, [src] + headers):
            jobs.append([hipcc] + FLAGS + ["-c", src, "-o", obj])

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        if verbose and r.stderr:
            print(r.stderr)

    with ThreadPoolExecutor(max_workers=4) as ex:
        list(ex.map(run, jobs))
    if force or jobs or _stale(LIB_PATH, objs):
        run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs)
    build_hostext(force)
    return LIB_PATH


def build_hostext(force: bool = False) -> str:
    """``_wwhostext.so`` - host code only (gcc, the interpreter's own headers); no GPU code in it."""
    import sysconfig
    src = os.path.join(CSRC, "hostext.c")
    if force or _stale(HOSTEXT_PATH, [src]):
        cc = shutil.which("gcc") or shutil.which("cc")
        if cc is None:
            raise RuntimeError("no C compiler: _wwhostext.so cannot be built")
        r = subprocess.run([cc, "-O2", "-shared", "-fPIC", "-Wall", "-I" + sysconfig.get_paths()["include"], src, "-o", HOSTEXT_PATH],
                           capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("building _wwhostext.so failed:\n" + r.stdout + r.stderr)
    return HOSTEXT_PATH


if __name__ == "__main__":
    print(build(verbose=True))
