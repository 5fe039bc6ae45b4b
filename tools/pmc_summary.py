#!/usr/bin/env python3
"""Aggregate rocprofv3 --pmc CSV passes per kernel: sum and per-dispatch mean of every counter."""
import csv, glob, os, sys, collections
out = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
ndisp = collections.defaultdict(lambda: collections.defaultdict(set))
for f in sorted(glob.glob(os.path.join(out, "p*", "**", "*counter_collection.csv"), recursive=True)):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:40]
        agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
        ndisp[k][r["Counter_Name"]].add(r["Dispatch_Id"])
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        n = max(1, len(ndisp[k][c]))
        print(f"    {c:32s} total {agg[k][c]:.6g}  per-dispatch {agg[k][c] / n:.6g}  (n={n})")
