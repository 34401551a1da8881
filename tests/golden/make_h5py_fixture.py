#!/usr/bin/env python
"""Writes tests/golden/h5py_features.h5 with REAL h5py / libhdf5 and the reference's exact calls
(/root/reference/utils/filter_dataset_to_h5.py:136-145: one `create_dataset(name, data=features)` per clip + the four
attributes), and tests/golden/h5py_features.npz with what went in.  Run it with an interpreter that has h5py - in the build
image:  /opt/conda/bin/python3.9 tests/golden/make_h5py_fixture.py   (h5py 3.3.0 on libhdf5 1.10.6).
wwhip.h5min must read the file bit for bit (tests/test_h5min.py::test_h5min_reads_a_file_written_by_h5py)."""
import os
import sys

import h5py
import numpy as np

here = os.path.dirname(os.path.abspath(__file__))
rng = np.random.default_rng(20261004)
clips = []
for i in range(12):
    rows = [0, 1, 37, 147, 182, 60][i % 6]
    clips.append({"file_name": "%08x-%04x" % (int(rng.integers(1 << 31)), i),   # hey-snips stems are hex ids
                  "features": rng.normal(3.0, 1.5, (rows, 40)).astype(np.float32),
                  "is_hotword": int(i % 5 == 0), "speaker": int(rng.integers(0, 300)),
                  "speech_start_ts": -1 if i % 7 == 0 else int(rng.integers(0, 40)),
                  "speech_end_ts": -1 if i % 7 == 0 else int(rng.integers(40, 250))})
path = os.path.join(here, "h5py_features.h5")
with h5py.File(path, "w") as h5f:
    for audio in clips:
        dset = h5f.create_dataset(audio["file_name"], data=audio["features"])
        dset.attrs["is_hotword"] = audio["is_hotword"]
        dset.attrs["speaker"] = audio["speaker"]
        dset.attrs["speech_start_ts"] = audio["speech_start_ts"]
        dset.attrs["speech_end_ts"] = audio["speech_end_ts"]
out = {"names": np.array([c["file_name"] for c in clips])}
for c in clips:
    out["f_" + c["file_name"]] = c["features"]
    out["a_" + c["file_name"]] = np.array([c["is_hotword"], c["speaker"], c["speech_start_ts"], c["speech_end_ts"]], np.int64)
np.savez(os.path.join(here, "h5py_features.npz"), **out)
print("wrote", path, os.path.getsize(path), "bytes; h5py", h5py.__version__, "libhdf5", h5py.version.hdf5_version, file=sys.stderr)
