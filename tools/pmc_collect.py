#!/usr/bin/env python3
"""rocprofv3 --pmc output directories -> profiles/r04/pmc_counters.json (what bench.py's roofline.traffic and
step_datapath_busy read).  Usage:

    python tools/pmc_collect.py OUT.json workload=DIR [workload=DIR ...]

Every DIR (and its sub-directories) is searched for *counter_collection.csv; per workload and kernel the MEAN value per
launch of every counter is stored, under the kernel names the library's own profile uses (logmel_kernel<f64>,
crnn_fused_kernel, ...).  The file carries the hash of the kernel sources it was measured on (bench.kernel_source_sha):
bench.py ignores a file measured on other sources."""
import collections, csv, glob, json, os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def short(name: str) -> str:
    n = name.replace("void ", "")
    m = re.match(r"(\w+)(<[^>]*>)?\(", n)
    base, targs = (m.group(1), m.group(2) or "") if m else (n, "")
    if base == "logmel_rows_kernel":  # round 4's fp64 front end (its own label in the library's profile since round 5)
        return "logmel_rows_kernel"
    if base == "logmel_kernel":
        return "logmel_kernel<f64>" if targs.startswith("<double") else "logmel_kernel<f32>"
    if base == "crnn_fused_kernel":
        return "crnn_fused_kernel<front>" if targs.startswith("<true") else "crnn_fused_kernel"
    if base == "crnn_fused_bf16_kernel":
        return "crnn_fused_kernel<front,bf16x3>" if targs.startswith("<true") else "crnn_fused_kernel<bf16x3>"
    if base == "wavenet_kernel":
        return "wavenet_kernel<bf16x3>" if re.match(r"<\w+, true", targs) else "wavenet_kernel"
    if base == "stream_frontend_kernel":
        return "stream_frontend_kernel"
    if base == "crnn_stream_kernel":  # <0>: behind a front-end kernel of its own; <1 | 2>: ONE launch per tick (round 5)
        return "crnn_stream_kernel" if targs.startswith("<0") else "crnn_stream_kernel<tick>"
    return base


def main():
    import bench
    out = {"source_sha": bench.kernel_source_sha(),
           "note": "rocprofv3 --pmc passes (tools/r06_prof.sh), mean per launch.  FETCH_SIZE / WRITE_SIZE in KiB as reported "
                   "(gfx950: FETCH_SIZE counts half of 16-byte-per-lane streaming reads -> traffic = 2 x FETCH + WRITE, "
                   "MI355X_MICROARCH.md); SQ_VALU_MFMA_BUSY_CYCLES in cycles summed over SIMDs; SQ_ACTIVE_INST_* / SQ_WAVE_CYCLES / "
                   "SQ_WAIT_* in quad-cycles summed over waves (same guide, s_memtime row).",
           "workloads": {}}
    for arg in sys.argv[2:]:
        wl, d = arg.split("=", 1)
        agg = collections.defaultdict(lambda: collections.defaultdict(float))
        cnt = collections.defaultdict(collections.Counter)
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r["Kernel_Name"])
                if "rocclr" in k or k.startswith("at::") or "elementwise" in k or "Cijk" in k:
                    continue
                agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
                cnt[k][r["Counter_Name"]] += 1
        rec = out["workloads"].setdefault(wl, {})
        for k, v in agg.items():
            rec.setdefault(k, {}).update({c: x / cnt[k][c] for c, x in v.items()})
            rec[k]["launches_averaged"] = int(max(cnt[k].values()))
    os.makedirs(os.path.dirname(os.path.abspath(sys.argv[1])), exist_ok=True)
    json.dump(out, open(sys.argv[1], "w"), indent=1, sort_keys=True)
    print(json.dumps({wl: sorted(v) for wl, v in out["workloads"].items()}))


if __name__ == "__main__":
    main()
