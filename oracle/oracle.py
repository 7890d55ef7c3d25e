"""ctypes binding for the CPU oracle (oracle/build/libmatchy_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg.
The product package (matchy_amd) never imports this module.
"""
import ctypes as C
import json
import os
import subprocess
from pathlib import Path

HERE = Path(__file__).resolve().parent
LIB = HERE / "build" / "libmatchy_oracle.so"
PSL = HERE.parent / "matchy_amd" / "data" / "psl.bin"

TYPE_NAMES = ["Domain", "Email", "IPv4", "IPv6", "MD5", "SHA1", "SHA256", "SHA384", "SHA512", "Bitcoin", "Ethereum", "Monero"]
EX_ALL = 255


class _Match(C.Structure):
    _fields_ = [("type", C.c_uint8), ("ip", C.c_uint8 * 16), ("start", C.c_uint64), ("end", C.c_uint64)]


class _Hit(C.Structure):
    _fields_ = [("start", C.c_uint64), ("end", C.c_uint64), ("type", C.c_uint8), ("kind", C.c_uint8),
                ("prefix_len", C.c_uint8), ("pad", C.c_uint8), ("ip_data_offset", C.c_uint32), ("n_ids", C.c_uint32)]


class ScanStats(C.Structure):
    _fields_ = [("lines", C.c_uint64), ("candidates", C.c_uint64), ("matches", C.c_uint64), ("bytes", C.c_uint64),
                ("by_type", C.c_uint64 * 12), ("seconds", C.c_double), ("chunks", C.c_uint64)]


def build(force=False):
    """Compile the oracle with g++ (oracle/Makefile)."""
    srcs = [HERE / n for n in ("matchy_oracle.cpp", "extractor.h", "database.h", "crypto.h", "Makefile")]
    if force or not LIB.exists() or any(s.stat().st_mtime > LIB.stat().st_mtime for s in srcs):
        subprocess.run(["make", "-C", str(HERE)], check=True, stdout=subprocess.DEVNULL)
    return LIB


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not LIB.exists():
            build()
        L = C.CDLL(str(LIB))
        L.orc_init.argtypes = [C.c_char_p]
        L.orc_to_lowercase.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.orc_to_lowercase.restype = C.c_size_t
        L.orc_psl_count.restype = C.c_size_t
        L.orc_psl_contains.argtypes = [C.c_char_p, C.c_size_t]
        L.orc_xxh64.argtypes = [C.c_char_p, C.c_size_t, C.c_uint64]
        L.orc_xxh64.restype = C.c_uint64
        for f in (L.orc_sha256, L.orc_keccak256):
            f.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.orc_extract_chunk.argtypes = [C.c_uint32, C.c_uint32, C.c_char_p, C.c_size_t, C.POINTER(_Match), C.c_size_t]
        L.orc_extract_chunk.restype = C.c_size_t
        L.orc_format_ip.argtypes = [C.c_int, C.c_char_p, C.c_char_p, C.c_size_t]
        L.orc_format_ip.restype = C.c_size_t
        L.orc_parse_ipv6.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p]
        L.orc_db_open.argtypes = [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.orc_db_open.restype = C.c_void_p
        L.orc_db_close.argtypes = [C.c_void_p]
        L.orc_db_lookup_json.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]
        L.orc_db_lookup_json.restype = C.c_size_t
        L.orc_db_metadata_json.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t]
        L.orc_db_metadata_json.restype = C.c_size_t
        L.orc_scan.argtypes = [C.c_void_p, C.c_char_p, C.c_size_t, C.c_uint32, C.c_size_t, C.c_int, C.c_size_t, C.c_char_p,
                               C.c_int, C.POINTER(ScanStats)]
        L.orc_scan.restype = C.c_void_p
        L.orc_scan_hit_count.argtypes = [C.c_void_p]
        L.orc_scan_hit_count.restype = C.c_size_t
        L.orc_scan_hit.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(_Hit), C.POINTER(C.c_uint32), C.POINTER(C.c_int64), C.c_size_t]
        L.orc_scan_ndjson.argtypes = [C.c_void_p, C.POINTER(C.c_size_t)]
        L.orc_scan_ndjson.restype = C.c_void_p
        L.orc_scan_free.argtypes = [C.c_void_p]
        # MATCHY_AMD_PSL: the same override the product honours (tests with a public-suffix list of their own)
        psl = os.environ.get("MATCHY_AMD_PSL") or str(PSL)
        if L.orc_init(psl.encode()) != 0:
            raise RuntimeError(f"oracle: cannot load PSL container {psl}")
        _lib = L
    return _lib


def xxh64(data: bytes, seed=0) -> int:
    return lib().orc_xxh64(data, len(data), seed)


def sha256(data: bytes) -> bytes:
    out = C.create_string_buffer(32)
    lib().orc_sha256(data, len(data), out)
    return out.raw


def keccak256(data: bytes) -> bytes:
    out = C.create_string_buffer(32)
    lib().orc_keccak256(data, len(data), out)
    return out.raw


def format_ip(ip: bytes, v6: bool) -> str:
    buf = C.create_string_buffer(64)
    lib().orc_format_ip(1 if v6 else 0, bytes(ip), buf, 64)
    return buf.value.decode()


def to_lowercase(text: str) -> str:
    """Rust str::to_lowercase as restated in oracle/unicode_lower.h."""
    b = text.encode("utf-8")
    out = C.create_string_buffer(len(b) * 3 + 8)
    n = lib().orc_to_lowercase(b, len(b), out, len(out))
    return out.raw[:n].decode("utf-8")


def extract(data: bytes, flags=EX_ALL, min_labels=2):
    """Reference chunk-path extraction → list of (type_name, start, end, canonical_value)."""
    L = lib()
    cap = max(64, len(data) // 4 + 16)
    arr = (_Match * cap)()
    n = L.orc_extract_chunk(flags, min_labels, data, len(data), arr, cap)
    if n > cap:
        arr = (_Match * n)()
        n = L.orc_extract_chunk(flags, min_labels, data, len(data), arr, n)
    out = []
    for i in range(n):
        m = arr[i]
        t = TYPE_NAMES[m.type]
        if t == "IPv4":
            val = format_ip(bytes(m.ip[:4]), False)
        elif t == "IPv6":
            val = format_ip(bytes(m.ip), True)
        else:
            val = data[m.start:m.end].decode("utf-8")
        out.append((t, m.start, m.end, val))
    return out


class Database:
    def __init__(self, blob: bytes):
        err = C.create_string_buffer(512)
        self._blob = bytes(blob)
        self._h = lib().orc_db_open(self._blob, len(self._blob), err, 512)
        if not self._h:
            raise ValueError("oracle db open failed: " + err.value.decode())

    def close(self):
        if self._h:
            lib().orc_db_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def lookup(self, query) -> dict:
        q = query.encode() if isinstance(query, str) else bytes(query)
        cap = 1 << 16
        buf = C.create_string_buffer(cap)
        n = lib().orc_db_lookup_json(self._h, q, len(q), buf, cap)
        if n + 1 > cap:
            buf = C.create_string_buffer(n + 1)
            lib().orc_db_lookup_json(self._h, q, len(q), buf, n + 1)
        return json.loads(buf.value.decode())

    def metadata(self) -> dict:
        cap = 1 << 16
        buf = C.create_string_buffer(cap)
        lib().orc_db_metadata_json(self._h, buf, cap)
        return json.loads(buf.value.decode())

    def scan(self, data: bytes, flags=0, chunk_bytes=0, threads=1, cache=0, source="-", want_json=True):
        """Reference parallel-path scan. Returns (hits, ndjson_lines, stats).
        hits: list of dict(start,end,type,kind,prefix_len,ip_data_offset,ids,offs) in canonical order."""
        L = lib()
        st = ScanStats()
        r = L.orc_scan(self._h, data, len(data), flags, chunk_bytes, threads, cache, source.encode(), 1 if want_json else 0, C.byref(st))
        try:
            n = L.orc_scan_hit_count(r)
            hits = []
            h = _Hit()
            ids = (C.c_uint32 * 256)()
            offs = (C.c_int64 * 256)()
            for i in range(n):
                L.orc_scan_hit(r, i, C.byref(h), ids, offs, 256)
                k = min(h.n_ids, 256)
                hits.append(dict(start=h.start, end=h.end, type=TYPE_NAMES[h.type], kind={2: "ip", 3: "pattern"}[h.kind],
                                 prefix_len=h.prefix_len, ip_data_offset=h.ip_data_offset,
                                 ids=list(ids[:k]), offs=list(offs[:k])))
            ln = C.c_size_t()
            p = L.orc_scan_ndjson(r, C.byref(ln))
            nd = C.string_at(p, ln.value).decode("utf-8") if ln.value else ""
            lines = nd.split("\n")[:-1] if nd else []
        finally:
            L.orc_scan_free(r)
        return hits, lines, st
