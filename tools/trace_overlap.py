#!/usr/bin/env python3
"""Development tool: from a rocprofv3 kernel_trace.csv, how many kernels run at the same time (share of the wall
time with 0, 1, 2, ... kernels in flight) over the densest 20 ms of the trace."""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
ev = []
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    ev.append((s, 1, r["Kernel_Name"][:24])); ev.append((e, -1, r["Kernel_Name"][:24]))
ev.sort()
t_end = ev[-1][0]
t_lo = t_end - int(float(sys.argv[2]) * 1e6) if len(sys.argv) > 2 else t_end - 20_000_000
hist = collections.Counter(); cur = 0; last = None
for t, d, _ in ev:
    if last is not None and t > t_lo:
        hist[cur] += t - max(last, t_lo)
    cur += d; last = t
tot = sum(hist.values())
print({k: round(v / tot, 3) for k, v in sorted(hist.items())})
