#!/usr/bin/env python3
"""Development tool: rocprofv3 kernel_trace.csv of a pipelined run -> per-kernel mean duration over the last N launches,
the share of wall time with k kernels in flight, and a text timeline of a few steps.  usage: trace_timeline.py csv [n_last]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 400
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-n_last:]
dur = collections.defaultdict(list)
for r in rows:
    dur[r["Kernel_Name"].split("(")[0][:40]].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in dur.items():
    v.sort()
    print(f"{k:42s} n={len(v):4d} mean={sum(v)/len(v)/1e3:7.2f} us  p10={v[len(v)//10]/1e3:7.2f} p50={v[len(v)//2]/1e3:7.2f} p90={v[len(v)*9//10]/1e3:7.2f}")
t0 = int(rows[0]["Start_Timestamp"]); t1 = max(int(r["End_Timestamp"]) for r in rows)
print(f"wall {(t1-t0)/1e3:.1f} us for {len(rows)} launches -> {(t1-t0)/1e3/(len(rows)/2):.2f} us per step (2 launches per step)")
ev = []
for r in rows:
    kind = "l" if "logmel" in r["Kernel_Name"] or "frontend" in r["Kernel_Name"] else "c"  # (names start with "void ...": classify by what they contain)
    ev.append((int(r["Start_Timestamp"]), 1, kind)); ev.append((int(r["End_Timestamp"]), -1, kind))
ev.sort()
hist = collections.Counter(); cur = collections.Counter(); last = None
for t, d_, nm in ev:
    if last is not None:
        hist[(cur["l"], cur["c"])] += t - last
    cur[nm] += d_; last = t
tot = sum(hist.values())
print("share of wall time by (front-end kernels, crnn kernels) in flight:")
for k, v in sorted(hist.items()):
    print("  ", k, round(v / tot, 3))
print("timeline of the last 16 launches (us since the first of them): queue start end name")
base = int(rows[-16]["Start_Timestamp"])
for r in rows[-16:]:
    print(f"  q{r.get('Queue_Id','?'):>3s} {(int(r['Start_Timestamp'])-base)/1e3:8.2f} {(int(r['End_Timestamp'])-base)/1e3:8.2f}  {r['Kernel_Name'][:30]}")
