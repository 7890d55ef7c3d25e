"""Synthetic workloads of BASELINE.json for tests and bench.py — not part of the product library.

`make_log` / `ioc_keys` wrap tools/synthgen.cpp (built on demand with g++); `build_db` feeds the indicator set
through the product's own builder C ABI (matchy_builder_add), i.e. the same path `matchy build` uses.
"""
import ctypes as C
import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent
LIB = HERE / "build" / "libsynthgen.so"
SRC = HERE / "synthgen.cpp"

SEED = 0x6D61746368790001  # "matchy\0\1" — SURVEY §8d


class Cfg(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("n_ip", C.c_uint32), ("n_cidr", C.c_uint32), ("n_dom", C.c_uint32),
                ("n_hash", C.c_uint32), ("n_glob", C.c_uint32), ("hit_permille", C.c_uint32), ("cidr_mode", C.c_uint32)]


def config(name: str) -> Cfg:
    """Indicator mixes of the BASELINE configs (scaled variants carry a suffix, e.g. 'c2/100')."""
    scale = 1
    if "/" in name:
        name, s = name.split("/")
        scale = int(s)
    glob_prefix = name == "c3b"   # C3 "AC-forced" variant: keys written glob:{key} -> 1M AC literals
    if glob_prefix:
        name = "c3"
    presets = {
        # C1: 1K-indicator CSV (400 /32, 100 CIDR, 350 domains, 100 globs, 50 hashes)
        "c1": dict(n_ip=400, n_cidr=100, n_dom=350, n_hash=50, n_glob=100),
        # C2: 100K mixed IoCs, no globs
        "c2": dict(n_ip=40000, n_cidr=10000, n_dom=35000, n_hash=15000, n_glob=0),
        # C3: 1M domain/hash indicators
        "c3": dict(n_ip=0, n_cidr=0, n_dom=700000, n_hash=300000, n_glob=0),
        # C4: C2 with 10K domains replaced by *.domain globs
        "c4": dict(n_ip=40000, n_cidr=10000, n_dom=25000, n_hash=15000, n_glob=10000),
        # C5: 10M IoCs, CIDR-heavy ip-trie (9M: 70 % /24, 20 % /16../23, 10 % /32) + 1M domains
        "c5": dict(n_ip=0, n_cidr=9000000, n_dom=1000000, n_hash=0, n_glob=0),
    }
    p = {k: max(v // scale, 1 if v else 0) for k, v in presets[name].items()}
    cfg = Cfg(seed=SEED, hit_permille=20, cidr_mode=1 if name == "c5" else 0, **p)
    cfg.glob_prefix = glob_prefix
    return cfg


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not LIB.exists() or LIB.stat().st_mtime < SRC.stat().st_mtime:
            # several ranks of one node may get here at once: build under a lock, into a private file, rename into place
            import fcntl
            import os
            LIB.parent.mkdir(exist_ok=True)
            with open(LIB.parent / ".synthgen.lock", "w") as lk:
                fcntl.flock(lk, fcntl.LOCK_EX)
                if not LIB.exists() or LIB.stat().st_mtime < SRC.stat().st_mtime:
                    tmp = LIB.with_suffix(f".{os.getpid()}.tmp")
                    subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-o", str(tmp), str(SRC)], check=True)
                    os.replace(tmp, LIB)
        L = C.CDLL(str(LIB))
        L.synth_log.argtypes = [C.POINTER(Cfg), C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t]
        L.synth_log.restype = C.c_size_t
        L.synth_log_shape.argtypes = [C.POINTER(Cfg), C.c_int, C.c_uint64, C.c_uint64, C.c_uint64, C.c_void_p, C.c_size_t]
        L.synth_log_shape.restype = C.c_size_t
        L.synth_ioc_feed.argtypes = [C.POINTER(Cfg), C.c_int, C.c_void_p, C.c_void_p]
        L.synth_ioc_feed.restype = C.c_longlong
        for f in (L.synth_ioc_key, L.synth_ioc_data):
            f.argtypes = [C.POINTER(Cfg), C.c_int, C.c_uint32, C.c_char_p, C.c_size_t]
            f.restype = C.c_size_t
        _lib = L
    return _lib


# log shapes (tools/synthgen.cpp gen_shape): name -> id; "skewed-halves" takes the period in lines as `shape_param`
SHAPES = {"nginx": 0, "jsonl-app": 1, "ip-dense": 2, "url-heavy": 3, "hash-dense": 4, "skewed-halves": 5}


def make_log(cfg: Cfg, first_line: int, n_lines: int, shape: str = "nginx", shape_param: int = 0) -> bytes:
    L = lib()
    sid = SHAPES[shape]
    cap = n_lines * 360 + 1024
    buf = C.create_string_buffer(cap)
    n = L.synth_log_shape(C.byref(cfg), sid, shape_param, first_line, n_lines, buf, cap)
    if n > cap:
        buf = C.create_string_buffer(n)
        n = L.synth_log_shape(C.byref(cfg), sid, shape_param, first_line, n_lines, buf, n)
    return buf.raw[:n]


def make_log_into(cfg: Cfg, first_line: int, n_lines: int, ptr: int, cap: int, shape: str = "nginx", shape_param: int = 0) -> int:
    """Generate straight into caller memory (e.g. a pinned torch tensor). Returns bytes written (or needed)."""
    return lib().synth_log_shape(C.byref(cfg), SHAPES[shape], shape_param, first_line, n_lines, ptr, cap)


KINDS = (("ip", 0, "n_ip"), ("cidr", 1, "n_cidr"), ("domain", 2, "n_dom"), ("hash", 3, "n_hash"), ("glob", 4, "n_glob"))


def ioc_entries(cfg: Cfg):
    """Yield (key, json_data) for every indicator of the config, in CSV row order."""
    L = lib()
    kb = C.create_string_buffer(512)
    db = C.create_string_buffer(512)
    for _, kind, field in KINDS:
        for i in range(getattr(cfg, field)):
            L.synth_ioc_key(C.byref(cfg), kind, i, kb, 512)
            L.synth_ioc_data(C.byref(cfg), kind, i, db, 512)
            key = kb.value
            if getattr(cfg, "glob_prefix", False) and kind >= 2:
                key = b"glob:" + key   # substring semantics through the paraglob section (Q9)
            yield key, db.value


def build_db(cfg: Cfg, epoch=1700000000, case_insensitive=False) -> bytes:
    import matchy_amd as M
    ML = M.lib()
    b = M.DatabaseBuilder(build_epoch=epoch, case_insensitive=case_insensitive)
    # the whole feed runs in C++ (tools/synthgen.cpp calls matchy_builder_add directly: 10M entries for C5)
    fed = lib().synth_ioc_feed(C.byref(cfg), 1 if getattr(cfg, "glob_prefix", False) else 0,
                               C.cast(ML.matchy_builder_add, C.c_void_p), b._h)
    if fed < 0:
        raise ValueError(f"builder rejected entry #{-fed - 1}: {M.last_error()}")
    blob = b.build()
    b.close()
    return blob
