"""Times the host side of Scanner.scan_device for the bench workload (diagnostic)."""
import sys, time
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import matchy_amd as M
from tools import synth

lines = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
cfg = synth.config("c2")
blob = synth.build_db(cfg)
db = M.Database(blob)
sc = M.Scanner(db, profile=True)
cap = lines * 200 + (1 << 20)
host = torch.empty(cap, dtype=torch.uint8)
n = synth.make_log_into(cfg, 0, lines, host.data_ptr(), cap)
d = torch.empty(n + 64, dtype=torch.uint8, device="cuda")
d[:n].copy_(host[:n])
torch.cuda.synchronize()
st = torch.cuda.current_stream().cuda_stream
for mode in (0, 1, 3, 0, 1):
    for rep in range(3):
        t0 = time.perf_counter()
        r = sc.scan_device(d.data_ptr(), n, stream=st, fetch_mode=mode)
        t1 = time.perf_counter()
        r.close()
        t2 = time.perf_counter()
        print(f"mode={mode} scan_device={1e3*(t1-t0):8.3f} ms close={1e3*(t2-t1):7.3f} ms kernels={sc.timing_ms()['total']:.3f} ms hits={r.n_hits}")
