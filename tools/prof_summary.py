#!/usr/bin/env python3
"""Summarise a tools/prof.sh output directory: per-kernel time (rocprofv3 --stats) and mean PMC counters per dispatch."""
import collections
import csv
import glob
import sys

out = sys.argv[1]
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print("== rocprofv3 --kernel-trace --stats:", f.split("/")[-1])
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:60]:60s} calls={r['Calls']:>4s} avg_ns={float(r['AverageNs']):12.0f} pct={r['Percentage']}")
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0][:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== mean PMC counter values per dispatch")
for k, v in sorted(agg.items()):
    if k.startswith("mxy::"):
        print(k)
        for c, x in sorted(v.items()):
            print(f"    {c:28s} {sum(x) / len(x):18.0f}")
