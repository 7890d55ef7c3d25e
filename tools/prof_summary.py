#!/usr/bin/env python3
"""Summarise a tools/prof.sh output directory: per-kernel time (rocprofv3 --stats), mean PMC counters per dispatch,
and the HBM traffic per launch corrected as /opt/skills/guides/MI355X_MICROARCH.md (§HBM) prescribes:
FETCH_SIZE and WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-B requests at 64 B, so it is doubled;
WRITE_SIZE is exact. Writes <dir>/traffic.json next to the printed summary."""
import collections
import csv
import glob
import json
import sys

out = sys.argv[1]
times = {}
for f in glob.glob(out + "/trace/**/*kernel_stats.csv", recursive=True):
    print("== rocprofv3 --kernel-trace --stats:", f.split("/")[-1])
    for r in csv.DictReader(open(f)):
        print(f"{r['Name'][:60]:60s} calls={r['Calls']:>4s} avg_ns={float(r['AverageNs']):12.0f} pct={r['Percentage']}")
        times[r["Name"].split("(")[0].replace("void ", "")] = float(r["AverageNs"])
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(out + "/pmc*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")[:40]][r["Counter_Name"]].append(float(r["Counter_Value"]))
print("== mean PMC counter values per dispatch")
traffic = {}
for k, v in sorted(agg.items()):
    if k.startswith("mxy::"):
        print(k)
        for c, x in sorted(v.items()):
            print(f"    {c:28s} {sum(x) / len(x):18.0f}")
        mean = {c: sum(x) / len(x) for c, x in v.items()}
        if "FETCH_SIZE" in mean:
            rd = mean["FETCH_SIZE"] * 1024 * 2
            wr = mean.get("WRITE_SIZE", 0.0) * 1024
            traffic[k] = {"fetch_size_kib": mean["FETCH_SIZE"], "write_size_kib": mean.get("WRITE_SIZE"),
                          "hbm_read_bytes": rd, "hbm_write_bytes": wr, "hbm_bytes": rd + wr,
                          "avg_ns": times.get(k)}
print("== HBM traffic per launch (FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024)")
for k, t in traffic.items():
    print(f"{k:40s} read {t['hbm_read_bytes'] / 1e9:8.3f} GB  write {t['hbm_write_bytes'] / 1e9:8.3f} GB")
nbytes = None
try:
    for line in open(out + "/trace.log"):
        if line.startswith("{") and "bytes_per_gpu" in line:
            nbytes = json.loads(line)["config"]["bytes_per_gpu"]
except OSError:
    pass
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.abspath(__file__)))
from csrc_hash import csrc_sha
json.dump({"bytes_per_gpu": nbytes, "csrc_sha": csrc_sha(), "kernels": traffic}, open(out + "/traffic.json", "w"), indent=1)
