#!/usr/bin/env python3
"""Development (round 6): from a rocprofv3 kernel trace of `tools/eval_share.py N 1 only8` - per evaluation pass (kernels separated
by idle gaps > 400 us belong to different passes): the span from the first kernel's start to the last one's end, the time with at
least one kernel in flight, and the idle gaps inside the span (the GPU waiting for the host).  usage: eval_pass_trace.py kernel_trace.csv"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if any(k in r["Kernel_Name"] for k in ("logmel", "crnn", "gru_tail", "pick_kernel", "sweep", "smooth"))]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
passes, cur, end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if end is not None and s - end > 400_000:
        passes.append(cur); cur = []
    cur.append((s, e, r["Kernel_Name"].split("(")[0].replace("void ", "")))
    end = max(end or 0, e)
passes.append(cur)
for g in passes[-6:]:
    t0 = g[0][0]
    span = max(e for _, e, _ in g) - t0
    ev = sorted([(s, 1) for s, e, _ in g] + [(e, -1) for s, e, _ in g])
    busy, depth, last, gaps = 0, 0, ev[0][0], []
    for t, d in ev:
        if depth > 0: busy += t - last
        elif t - last > 2000: gaps.append(((last - t0) / 1e3, (t - last) / 1e3))
        last = t; depth += d
    print(f"pass: {len(g)} kernels, span {span / 1e3:.0f} us, >= 1 kernel in flight {busy / 1e3:.0f} us, idle inside {(span - busy) / 1e3:.0f} us; gaps (at us: length us): "
          + ", ".join(f"{a:.0f}: {b:.0f}" for a, b in gaps))
    print("   " + " ".join(f"{n}@{(s - t0) / 1e3:.0f}+{(e - s) / 1e3:.0f}" for s, e, n in g))
