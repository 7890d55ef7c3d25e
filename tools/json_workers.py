"""`matchy match --format json` against `--format summary` for 1..4 scanners on one GPU and two batch sizes (file in the page cache)."""
import subprocess, sys, time, os
sys.path.insert(0, ".")
from tools import synth
cfg = synth.config("c2")
open("/tmp/c2.mxy", "wb").write(synth.build_db(cfg))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
with open("/tmp/c2.log", "wb") as f:
    for a in range(0, n, 1_000_000):
        f.write(synth.make_log(cfg, a, min(1_000_000, n - a)))
size = os.path.getsize("/tmp/c2.log")
cli = "matchy_amd/bin/matchy"
for fmt in ("summary", "json"):
    for devs in ("0", "0,0", "0,0,0", "0,0,0,0"):
        for bb in (64 << 20, 256 << 20):
            t = time.time()
            r = subprocess.run([cli, "match", "/tmp/c2.mxy"] + ["/tmp/c2.log"] * reps + ["--devices", devs, "--batch-bytes", str(bb), "--format", fmt, "-s"],
                               stdout=open("/dev/null", "wb"), stderr=subprocess.PIPE)
            dt = time.time() - t
            thr = [l.split("Throughput:")[1].strip() for l in r.stderr.decode().splitlines() if "Throughput" in l]
            print(f"format={fmt:8s} devices={devs:8s} batch={bb >> 20:4d}MiB: wall {dt:.2f}s = {size * reps / dt / 1e9:.2f} GB/s; scan phase {thr}", flush=True)
