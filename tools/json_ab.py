"""NDJSON output rate of `matchy match --format json` on the GPU box: python tools/json_ab.py (after tools/cli_fixed.py has written /tmp/c2.mxy and /tmp/c2.log).
JSON_AB_FILE=1 writes the output to a file instead of /dev/null; JSON_AB_OLD=<dir with an older matchy + libmatchy_amd.so> adds that build."""
import os, subprocess, sys, time
size = os.path.getsize("/tmp/c2.log")
reps = 10
to_file = bool(os.environ.get("JSON_AB_FILE"))
builds = [("this tree", "matchy_amd/bin/matchy", {})]
old = os.environ.get("JSON_AB_OLD")
if old:
    builds.insert(0, ("older build", old + "/matchy", {"LD_LIBRARY_PATH": os.path.abspath(old), "MATCHY_AMD_PSL": os.path.abspath("matchy_amd/data/psl.bin")}))
for name, cli, env in builds:
    for fmt, jobs in (("json", "auto"), ("json", "8"), ("summary", "auto")):
        dest = open("/tmp/out.ndjson", "wb") if to_file else subprocess.DEVNULL
        t = time.time()
        r = subprocess.run([cli, "match", "/tmp/c2.mxy"] + ["/tmp/c2.log"] * reps + ["-j", jobs, "--format", fmt, "-s"], stdout=dest, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        dt = time.time() - t
        thr = [l.split("] ")[1] for l in r.stderr.decode().splitlines() if "Throughput" in l or "Total matches" in l]
        where = "a file" if to_file else "/dev/null"
        print(f"{name}: --format {fmt} -j {jobs}, {reps} x {size} B, output to {where}: wall {dt:.2f} s = {size * reps / dt / 1e9:.2f} GB/s; {thr}", flush=True)
