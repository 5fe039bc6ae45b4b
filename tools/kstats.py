#!/usr/bin/env python3
"""Print the per-kernel rows of a rocprofv3 *_kernel_stats.csv compactly."""
import csv, sys
for f in sys.argv[1:]:
    print(f)
    for r in csv.DictReader(open(f)):
        if r["Name"].startswith(("k_", "void k_")):
            print("   %-22s calls %3s total %8.1f ms avg %8.2f ms" % (r["Name"].split("(")[0][:22], r["Calls"], float(r["TotalDurationNs"]) / 1e6, float(r["AverageNs"]) / 1e6))
