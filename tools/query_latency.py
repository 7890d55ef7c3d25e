#!/usr/bin/env python3
"""matchy_query latency on the C2 database (100 K indicators), host path against the kernel path.
GPU box, repo root:  python tools/query_latency.py  > profiles/rNN_query_latency.txt"""
import os
import random
import re
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from tools import synth  # noqa: E402


def main():
    cfg = synth.config("c2")
    blob = synth.build_db(cfg)
    db = "/tmp/c2_query.mxy"
    Path(db).write_bytes(blob)
    rng = random.Random(1)
    keys = [k.decode() for k, _ in synth.ioc_entries(cfg)]
    log = synth.make_log(cfg, 0, 40000)
    ips = [t.decode() for t in re.findall(rb"\b\d{1,3}\.\d{1,3}\.\d{1,3}\.\d{1,3}\b", log)]
    names = [t.decode() for t in re.findall(rb"[a-z0-9\-]+(?:\.[a-z0-9\-]+)+", log)]
    for kind, qs in (("ip addresses of the log (2 % in the database)", ips[:60000]),
                     ("host names of the log (1 % in the database)", names[:60000]),
                     ("database keys (all found)", rng.sample(keys, 60000))):
        qf = "/tmp/queries.txt"
        Path(qf).write_text("\n".join(qs) + "\n")
        exe = "/tmp/query_latency"
        subprocess.run(["g++", "-O2", "-std=c++17", "-I", str(ROOT / "include"), str(ROOT / "tools/ubench/query_latency.cpp"), "-o", exe,
                        "-L", str(ROOT / "matchy_amd/lib"), "-lmatchy_amd", "-lpthread", f"-Wl,-rpath,{ROOT / 'matchy_amd/lib'}"], check=True)
        for gpu in ("0", "1"):
            n = len(qs) if gpu == "0" else 3000   # the kernel path takes tens of microseconds per query
            Path(qf).write_text("\n".join(qs[:n]) + "\n")
            print(f"--- {kind}; {'lookup KERNELS (MATCHY_AMD_QUERY_ON_GPU=1)' if gpu == '1' else 'HOST path (default)'}", flush=True)
            env = dict(os.environ, MATCHY_AMD_QUERY_ON_GPU=gpu)
            subprocess.run([exe, db, qf, "8"], env=env, check=True)


if __name__ == "__main__":
    main()
