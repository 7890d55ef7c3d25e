"""`matchy match --format json`: one rendering thread against the default (pieces of a large result rendered side by side) — the two
outputs must be byte-identical; wall time and scan-phase rate of both, beside --format summary."""
import hashlib, subprocess, sys, time, os
sys.path.insert(0, ".")
from tools import synth
cfg = synth.config("c2")
open("/tmp/c2.mxy", "wb").write(synth.build_db(cfg))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
with open("/tmp/c2.log", "wb") as f:
    for a in range(0, n, 1_000_000):
        f.write(synth.make_log(cfg, a, min(1_000_000, n - a)))
size = os.path.getsize("/tmp/c2.log")
cli = os.environ.get("MATCHY_CLI", "matchy_amd/bin/matchy")
sums = {}
for devs in ("0", "0,0"):
    for name, fmt, env in (("summary", "summary", {}), ("json 1 thread", "json", {"MATCHY_AMD_JSON_THREADS": "1"}), ("json default", "json", {})):
        out = "/tmp/out_%s.ndjson" % name.replace(" ", "_")
        t = time.time()
        r = subprocess.run([cli, "match", "/tmp/c2.mxy"] + ["/tmp/c2.log"] * reps + ["--devices", devs, "--batch-bytes", str(256 << 20), "--format", fmt, "-s"],
                           stdout=open(out, "wb"), stderr=subprocess.PIPE, env=dict(os.environ, **env))
        dt = time.time() - t
        thr = [l.split("Throughput:")[1].strip() for l in r.stderr.decode().splitlines() if "Throughput" in l]
        h = hashlib.sha256(open(out, "rb").read()).hexdigest()[:16] if fmt == "json" else "-"
        sums.setdefault(devs, {})[name] = h
        print(f"devices={devs:4s} {name:14s}: wall {dt:.2f}s = {size * reps / dt / 1e9:.2f} GB/s; scan phase {thr}; output {os.path.getsize(out)} B sha256 {h}", flush=True)
for devs, d in sums.items():
    assert d["json 1 thread"] == d["json default"], (devs, d)
print("byte-identical: OK")
