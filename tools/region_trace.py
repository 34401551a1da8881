#!/usr/bin/env python3
"""Development: from a rocprofv3 kernel_trace.csv of `bench.py --steps 20 --warmup 5 --no-extra --no-cpu-baseline`, the anatomy of the
timed regions - for the last few 20-step regions (40 kernels between two idle gaps): span, kernel-busy share by number of kernels in
flight, the lead-in (first front end alone) and the drain (last CRNN kernels alone), and the launch order of the first kernels.
usage: region_trace.py kernel_trace.csv"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "logmel_rows" in r["Kernel_Name"] or "crnn_fused_kernel" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
# regions = runs of kernels separated by idle gaps > 25 us
regions, cur, end = [], [], None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    if end is not None and s - end > 25_000:
        regions.append(cur); cur = []
    cur.append((s, e, "F" if "logmel" in r["Kernel_Name"] else "C"))
    end = max(end or 0, e)
regions.append(cur)
regions = [g for g in regions if len(g) == 40]
def overlapped(g):  # some kernel starts before the previous one has ended: a region of the pipelined (multi-context) leg
    return sum(1 for a, b in zip(g, g[1:]) if b[0] < a[1] - 1000) > 5
regions = [g for g in regions if overlapped(g)][-6:]
for g in regions:
    t0, t1 = g[0][0], max(e for _, e, _ in g)
    ev = sorted([(s, 1, k) for s, e, k in g] + [(e, -1, k) for s, e, k in g])
    hist, curF, curC, last = {}, 0, 0, None
    for t, d, k in ev:
        if last is not None:
            hist[(curF, curC)] = hist.get((curF, curC), 0) + t - last
        if k == "F": curF += d
        else: curC += d
        last = t
    first_c = min(s for s, e, k in g if k == "C")
    last_f_end = max(e for s, e, k in g if k == "F")
    print(f"region span {(t1 - t0) / 1e3:7.1f} us = {(t1 - t0) / 20e3:5.2f} us/step | until the first CRNN kernel starts {(first_c - t0) / 1e3:5.1f} us | "
          f"after the last front end ends {(t1 - last_f_end) / 1e3:5.1f} us | in flight (front ends, CRNN): "
          + " ".join(f"{k}:{v / (t1 - t0):.2f}" for k, v in sorted(hist.items())))
g = regions[-1]
print("first 12 and last 8 kernels of the last region (start, end in us; F = front end, C = fused CRNN):")
for s, e, k in g[:12] + g[-8:]:
    print(f"   {k} {(s - g[0][0]) / 1e3:8.1f} {(e - g[0][0]) / 1e3:8.1f}")
