#!/usr/bin/env python3
"""Development (round 6): from a rocprofv3 kernel trace of tools/eval_at_scale.py - how much of the evaluation pass's GPU time has
kernels of TWO chunks in flight (the chunks go to alternating lanes = HIP streams).  usage: eval_lanes_trace.py kernel_trace.csv"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
per_q = collections.Counter()
for r in rows:
    name = r["Kernel_Name"]
    if not any(k in name for k in ("logmel", "crnn", "gru_tail", "pick_kernel")):
        continue
    s, e, q = int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", r.get("Stream_Id", "?"))
    ev.append((s, 1)); ev.append((e, -1)); per_q[q] += e - s
ev.sort()
busy = collections.Counter(); depth = 0; last = ev[0][0]
for t, d in ev:
    busy[depth] += t - last; last = t; depth += d
span = ev[-1][0] - ev[0][0]
print(f"{len(ev) // 2} kernels of the evaluation path over {span / 1e6:.3f} ms of trace (both passes, the gaps between them included)")
for k in sorted(busy):
    print(f"  {k} kernels in flight: {busy[k] / 1e6:8.3f} ms  ({busy[k] / span:.1%})")
print("  kernel time by queue:", {q: round(v / 1e6, 3) for q, v in per_q.items()})
