"""Build libmatchy_amd.so (HIP kernels + host pipeline + C ABI) for gfx950 with hipcc, in-tree."""
import os
import subprocess
import sys
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
LIBDIR = PKG / "lib"
LIB = LIBDIR / "libmatchy_amd.so"
BINDIR = PKG / "bin"
CLI = BINDIR / "matchy"
SOURCES = ["k_anchor.hip", "validate_kernels.hip", "lookup_kernels.hip", "sort_hits.hip", "ip_tables.hip", "engine.cpp", "db_image.cpp", "db_builder.cpp", "data_codec.cpp", "unicode_lower.cpp", "host_topology.cpp", "host_lookup.cpp", "capi.cpp"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# lookup_kernels.hip: keep the dynamically indexed private arrays of the glob matcher (result list, star stack: 80 dwords,
# touched a few times per text) in scratch instead of in vector registers — 96 VGPRs instead of 512 for k_lookup<true>.
EXTRA_FLAGS = {"lookup_kernels.hip": ["-mllvm", "-amdgpu-promote-alloca-to-vector-limit=64"]}


def needs_build():
    if not LIB.exists() or not CLI.exists():
        return True
    t = LIB.stat().st_mtime
    deps = list(CSRC.glob("*")) + [PKG.parent / "include" / "matchy_amd.h", Path(__file__)]
    return any(d.stat().st_mtime > t for d in deps)


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    LIBDIR.mkdir(exist_ok=True)
    objs = []
    flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc", "-Wall", "-Wno-unused-function",
             "-Wno-unused-result", "-DNDEBUG"]
    # throw-away instrumented builds (kernel experiments): MATCHY_AMD_CFLAGS="-DMXY_ANCHOR_DEBUG" python -m matchy_amd.build --force
    flags += os.environ.get("MATCHY_AMD_CFLAGS", "").split()
    procs = []
    for src in SOURCES:
        obj = LIBDIR / (src.rsplit(".", 1)[0] + ".o")
        extra = EXTRA_FLAGS.get(src, [])
        cmd = [HIPCC, *flags, *extra, "-x", "hip", "-c", str(CSRC / src), "-o", str(obj)]
        if verbose:
            print(" ".join(cmd), file=sys.stderr)
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
        objs.append(str(obj))
    failed = False
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            failed = True
            sys.stderr.write(f"--- hipcc {src} failed ---\n{out.decode(errors='replace')}\n")
        elif verbose and out:
            sys.stderr.write(out.decode(errors="replace"))
    if failed:
        raise RuntimeError("hipcc compilation failed")
    link = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *objs, "-ldl", "-lpthread"]
    subprocess.run(link, check=True)
    # the `matchy` command line (build / match) on top of the library
    BINDIR.mkdir(exist_ok=True)
    cli = ["g++", "-O2", "-std=c++17", "-Wall", "-D__HIP_PLATFORM_AMD__", "-I/opt/rocm/include", str(CSRC / "cli_main.cpp"), "-o", str(CLI),
           f"-L{LIBDIR}", "-lmatchy_amd", "-lz", "-lpthread", "-Wl,-rpath,$ORIGIN/../lib"]
    subprocess.run(cli, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
