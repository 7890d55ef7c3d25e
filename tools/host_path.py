"""PCIe-inclusive rate of the host-buffer entry (matchy_scanner_scan: pageable host memory -> H2D -> scan -> hits).
Reported in DESIGN.md §5; never bench.py's `value`. Usage on the GPU box: python tools/host_path.py [lines]"""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import matchy_amd as M
from tools import synth

lines = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
cfg = synth.config("c2")
db = M.Database(synth.build_db(cfg))
sc = M.Scanner(db)
cap = lines * 200 + (1 << 20)
host = torch.empty(cap, dtype=torch.uint8)
n = synth.make_log_into(cfg, 0, lines, host.data_ptr(), cap)
pinned = torch.empty(n, dtype=torch.uint8).pin_memory()
pinned.copy_(host[:n])
for name, buf in (("pageable", host), ("pinned", pinned)):
    for rep in range(3):
        t0 = time.perf_counter()
        r = sc.scan_ptr(buf.data_ptr(), n)
        dt = time.perf_counter() - t0
        print(f"{name:9s} rep {rep}: {n} B in {1e3 * dt:8.2f} ms = {n / dt / 1e9:7.2f} GB/s  lines={r.lines} hits={r.n_hits}")
        r.close()
