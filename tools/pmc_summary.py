#!/usr/bin/env python3
"""Summarise a rocprofv3 counter_collection.csv: mean counter value per kernel (development tool)."""
import csv, glob, collections, sys
f = (glob.glob(sys.argv[1] + "/*/*counter_collection.csv") + glob.glob(sys.argv[1] + "/*counter_collection.csv"))[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:48]; agg[k][r["Counter_Name"]] += float(r["Counter_Value"]); cnt[k][r["Counter_Name"]] += 1
for k, v in agg.items():
    if "rocclr" in k or "at::" in k: continue
    print(k, {c: round(x / cnt[k][c]) for c, x in v.items()})
