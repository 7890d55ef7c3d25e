"""Kernel timeline of one bench step from a rocprofv3 --kernel-trace run: python tools/timeline.py <dir> [step index]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_anchor" in r["Kernel_Name"]]
i0 = idx[int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2]
t0 = int(rows[i0]["Start_Timestamp"])
for r in rows[i0 - 1:]:
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    if s > 60_000_000 or (r is not rows[i0] and "k_anchor" in r["Kernel_Name"]):
        break
    print(f"{r['Kernel_Name'][:50]:50s} start {s / 1000:9.1f} us  end {e / 1000:9.1f} us  dur {(e - s) / 1000:8.1f} us  queue {r.get('Queue_Id', '')}")
