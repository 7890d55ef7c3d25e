"""Latency of the extractor on hostile inputs (very long labels / local parts / runs): one lane walks them alone.
Diagnostic: python tools/hostile_inputs.py on a GPU box."""
import sys, time
from pathlib import Path; sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import matchy_amd as M
ex = M.Extractor()
for name, buf in [("1MB label x4", (b"a" * 1_000_000 + b".com ") * 4), ("many dots 2MB", b"a." * 1_000_000 + b"com "), ("long local part email", b"x" * 2_000_000 + b"@a.com "),
                  ("hex 4MB", b"0123456789abcdef" * 250_000 + b" "), ("colons", b"a:" * 1_000_000)]:
    t = time.time(); r = ex.extract_from_chunk(buf); first = time.time() - t   # includes growing the work buffers to this size
    dt = 1e9
    for _ in range(3):   # the HIP runtime pins a pageable source buffer on one of its first uses (one call of 15-30 ms): best of three
        t = time.time(); r = ex.extract_from_chunk(buf); dt = min(dt, time.time() - t)
    print(f"{name:24s} {len(buf):9d} B -> {len(r)} items in {dt*1e3:8.1f} ms (first call {first*1e3:.1f} ms)", flush=True)
