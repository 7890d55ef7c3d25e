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
cli = os.environ.get("MATCHY_CLI", "matchy_amd/bin/matchy")
def run(files, jobs):
    t = time.time()
    r = subprocess.run([cli, "match", "/tmp/c2.mxy"] + files + ["-j", jobs, "--batch-bytes", str(256 << 20), "--format", "summary", "-s"],
                       stdout=subprocess.DEVNULL, stderr=subprocess.PIPE, env=dict(os.environ, MATCHY_AMD_TRACE="1"))
    dt = time.time() - t
    err = r.stderr.decode().splitlines()
    info = [l.split("] ")[1] for l in err if "Throughput" in l]
    ph = {}
    for l in err:   # phase marks of the command line (MATCHY_AMD_TRACE): database open, all batches done
        for key in ("database open after", "all batches done after"):
            if key in l:
                ph[key] = float(l.split(key)[1].split()[0])
    scan_s = (ph.get("all batches done after", 0) - ph.get("database open after", 0)) / 1e3
    nsc = [l.split("(")[1].split()[0] for l in err if "HIP devices" in l]
    return dt, info, scan_s, (nsc[0] if nsc else "?")
for i in range(3):
    dt, info, _, _ = run(["/tmp/small.log"], "auto")
    print(f"fixed cost (1 MB input): wall {dt:.3f} s {info}", flush=True)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 30
for jobs in ("1", "2", "auto", "6", "8"):
    dt, info, scan_s, nsc = run(["/tmp/c2.log"] * reps, jobs)
    print(f"{reps} x {size} B, -j {jobs} ({nsc} scanner(s) on one GPU): scan phase {scan_s:.2f} s = {size * reps / max(scan_s, 1e-9) / 1e9:.2f} GB/s; "
          f"wall {dt:.2f} s = {size * reps / dt / 1e9:.2f} GB/s incl. process start and database upload; {info}", flush=True)
