"""Fixed cost and steady-state rate of `matchy match --format summary` on the GPU box: python tools/cli_fixed.py [reps]"""
import os, subprocess, sys, time
sys.path.insert(0, ".")
from tools import synth
cfg = synth.config("c2")
open("/tmp/c2.mxy", "wb").write(synth.build_db(cfg))
if not os.path.exists("/tmp/c2.log"):
    with open("/tmp/c2.log", "wb") as f:
        for a in range(0, 10_000_000, 1_000_000):
            f.write(synth.make_log(cfg, a, 1_000_000))
open("/tmp/small.log", "wb").write(open("/tmp/c2.log", "rb").read(1_000_000))
size = os.path.getsize("/tmp/c2.log")
cli = "matchy_amd/bin/matchy"
def run(files, devs):
    t = time.time()
    r = subprocess.run([cli, "match", "/tmp/c2.mxy"] + files + ["--devices", devs, "--batch-bytes", str(256 << 20), "--format", "summary", "-s"],
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE)
    dt = time.time() - t
    info = [l.split("] ")[1] for l in r.stderr.decode().splitlines() if "Throughput" in l or "Processing time" in l]
    return dt, info
for i in range(3):
    dt, info = run(["/tmp/small.log"], "0")
    print(f"fixed cost (1 MB input): wall {dt:.3f} s {info}", flush=True)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for devs in ("0", "0,0", "0,0,0"):
    dt, info = run(["/tmp/c2.log"] * reps, devs)
    print(f"{reps} x {size} B, devices {devs}: wall {dt:.2f} s = {size * reps / dt / 1e9:.2f} GB/s incl. process start and database upload; {info}", flush=True)
