"""End-to-end timing of `matchy match` (file in the page cache -> GPU -> NDJSON / summary) for a few device lists."""
import subprocess, sys, time, os
sys.path.insert(0, ".")
from tools import synth
cfg = synth.config("c2")
open("/tmp/c2.mxy", "wb").write(synth.build_db(cfg))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 1     # the file is passed `reps` times (steady state past the start-up cost)
with open("/tmp/c2.log", "wb") as f:
    for a in range(0, n, 1_000_000):
        f.write(synth.make_log(cfg, a, min(1_000_000, n - a)))
size = os.path.getsize("/tmp/c2.log")
cli = "matchy_amd/bin/matchy"
for devs, bb, fmt in (("0", 256 << 20, "summary"), ("0,0", 256 << 20, "summary"), ("0,0", 64 << 20, "summary"), ("0,0,0", 64 << 20, "summary"),
                      ("0,0,0,0", 128 << 20, "summary"), ("0,0", 64 << 20, "json")):
    t = time.time()
    r = subprocess.run([cli, "match", "/tmp/c2.mxy"] + ["/tmp/c2.log"] * reps + ["--devices", devs, "--batch-bytes", str(bb), "--format", fmt, "-s"],
                       stdout=open("/tmp/out.ndjson", "wb"), stderr=subprocess.PIPE)
    dt = time.time() - t
    thr = [l for l in r.stderr.decode().splitlines() if "Throughput" in l or "Total matches" in l]
    print(f"devices={devs} batch={bb >> 20}MiB format={fmt}: wall {dt:.2f}s = {size * reps / dt / 1e9:.2f} GB/s incl. process start and DB upload; {thr}", flush=True)
