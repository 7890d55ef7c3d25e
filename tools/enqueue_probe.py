"""How long does the host take to enqueue one forked scan (all kernels and events of Scanner::scan_device)? matchy_scanner_submit_device
against submit + wait on the headline batch. A scan is not launch-bound when this is far below the duration of k_anchor."""
import ctypes, sys, time
sys.path.insert(0, ".")
import matchy_amd as M
from tools import synth
cfg = synth.config("c2")
blob = synth.build_db(cfg)
log = synth.make_log(cfg, 0, 10_000_000)
db = M.Database(blob); sc = M.Scanner(db)
hip = ctypes.CDLL("libamdhip64.so")
d = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(len(log) + 64)) == 0
assert hip.hipMemcpy(d, log, ctypes.c_size_t(len(log)), 1) == 0
for _ in range(5):
    r = sc.scan_device(d.value, len(log), fetch_mode=3); r.close()
L = M.lib()
ts = []
for _ in range(20):
    t0 = time.perf_counter()
    h = sc.submit_device(d.value, len(log), fetch_mode=3) if hasattr(sc, "submit_device") else None
    t1 = time.perf_counter()
    r = sc.wait() if h is not None or True else None
    t2 = time.perf_counter()
    ts.append(((t1 - t0) * 1e6, (t2 - t0) * 1e6))
    if r is not None: r.close()
print("submit (host enqueue of all launches) us:", sorted(round(a) for a, _ in ts)[len(ts)//2], " submit+wait us:", sorted(round(b) for _, b in ts)[len(ts)//2])
