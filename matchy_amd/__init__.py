"""matchy_amd — MI355X-native `matchy match` engine (host mirror over the C ABI in include/matchy_amd.h).

The classes mirror the reference's operator interface for this path:
  DatabaseBuilder  ~ matchy::DatabaseBuilder   (crates/matchy-format/src/mmdb_builder.rs)
  Database         ~ matchy::Database          (crates/matchy/src/database.rs)         — lookups run on the GPU
  Extractor        ~ matchy::extractor::Extractor (crates/matchy-extractor/src/lib.rs) — extraction runs on the GPU
  Scanner          ~ processing::Worker        (crates/matchy/src/processing/mod.rs:318-448), the bulk scan entry

There is no CPU fallback anywhere in this package: if libmatchy_amd.so is missing the import of the native
layer raises, and without a HIP device Database()/Extractor() raise RuntimeError.
"""
import ctypes as C
import os
import json
from pathlib import Path

from . import build as _build

__all__ = ["DatabaseBuilder", "Database", "Extractor", "Scanner", "lib", "ITEM_TYPE_NAMES", "EXTRACT_ALL", "last_error"]

PKG = Path(__file__).resolve().parent
LIB_PATH = PKG / "lib" / "libmatchy_amd.so"

ITEM_TYPE_NAMES = ["Domain", "Email", "IPv4", "IPv6", "MD5", "SHA1", "SHA256", "SHA384", "SHA512", "Bitcoin", "Ethereum", "Monero"]
EXTRACT_DOMAINS, EXTRACT_EMAILS, EXTRACT_IPV4, EXTRACT_IPV6 = 1, 2, 4, 8
EXTRACT_HASHES, EXTRACT_BITCOIN, EXTRACT_ETHEREUM, EXTRACT_MONERO = 16, 32, 64, 128
EXTRACT_ALL = 255

# every symbol include/matchy_amd.h declares
EXPORTED_SYMBOLS = [
    "matchy_builder_new", "matchy_builder_set_case_insensitive", "matchy_builder_add", "matchy_builder_set_description",
    "matchy_builder_save", "matchy_builder_build", "matchy_builder_free", "matchy_init_open_options",
    "matchy_open_with_options", "matchy_open", "matchy_open_buffer", "matchy_close", "matchy_query", "matchy_query_into",
    "matchy_free_result", "matchy_free_string", "matchy_result_to_json", "matchy_version", "matchy_format",
    "matchy_has_ip_data", "matchy_has_string_data", "matchy_has_literal_data", "matchy_has_glob_data", "matchy_metadata",
    "matchy_get_pattern_string", "matchy_pattern_count", "matchy_extractor_create", "matchy_extractor_extract_chunk",
    "matchy_matches_free", "matchy_extractor_free", "matchy_item_type_name", "matchy_scanner_create", "matchy_scanner_free",
    "matchy_scanner_scan", "matchy_scanner_scan_device", "matchy_scan_result_free", "matchy_scan_hit_to_json",
    "matchy_scanner_set_profile", "matchy_scanner_get_timing", "matchy_amd_last_error", "matchy_builder_set_build_epoch",
    "matchy_get_stats", "matchy_clear_cache", "matchy_has_pattern_data", "matchy_result_get_entry", "matchy_aget_value",
    "matchy_get_entry_data_list", "matchy_free_entry_data_list", "matchy_validate", "matchy_builder_set_schema",
    "matchy_amd_query_json", "matchy_amd_extractor_create", "matchy_amd_device_count", "matchy_scanner_submit_device",
    "matchy_scanner_wait", "matchy_scanner_set_slices", "matchy_scanner_last_slices", "matchy_scan_result_on_device",
    "matchy_amd_ac_dfa_states", "matchy_amd_suffix_filter", "matchy_amd_pinned_alloc", "matchy_amd_pinned_free",
    "matchy_amd_host_register", "matchy_amd_host_unregister",
    "matchy_amd_device_numa_node", "matchy_amd_bind_thread_to_device", "matchy_amd_numa_cpus",
    "matchy_multi_scanner_create", "matchy_multi_scanner_free", "matchy_multi_scanner_workers", "matchy_multi_scanner_worker_scanner",
    "matchy_multi_scanner_set_batch_hook", "matchy_multi_scanner_submit", "matchy_multi_scanner_next", "matchy_multi_scanner_scan",
    "matchy_multi_scanner_pending", "matchy_multi_scanner_max_pending", "matchy_multi_scanner_submit_near", "matchy_multi_scanner_worker_numa",
    "matchy_amd_unbind_thread",
    "matchy_multi_scanner_scan_file", "matchy_scan_result_to_ndjson",
]


class _Result(C.Structure):
    _fields_ = [("found", C.c_bool), ("prefix_len", C.c_uint8), ("_data_cache", C.c_void_p), ("_db_ref", C.c_void_p)]


class _Stats(C.Structure):
    _fields_ = [(n, C.c_uint64) for n in ("total_queries", "queries_with_match", "queries_without_match", "cache_hits",
                                          "cache_misses", "ip_queries", "string_queries")]


class _Entry(C.Structure):
    _fields_ = [("db", C.c_void_p), ("data_ptr", C.c_void_p)]


class _EntryValue(C.Union):
    _fields_ = [("pointer", C.c_uint32), ("utf8_string", C.c_char_p), ("double_value", C.c_double), ("bytes", C.POINTER(C.c_uint8)),
                ("uint16", C.c_uint16), ("uint32", C.c_uint32), ("int32", C.c_int32), ("uint64", C.c_uint64),
                ("uint128", C.c_uint8 * 16), ("boolean", C.c_bool), ("float_value", C.c_float)]


class _EntryData(C.Structure):
    _fields_ = [("has_data", C.c_bool), ("type_", C.c_uint32), ("value", _EntryValue), ("data_size", C.c_uint32), ("offset", C.c_uint32)]


class _EntryDataList(C.Structure):
    pass


_EntryDataList._fields_ = [("entry_data", _EntryData), ("next", C.POINTER(_EntryDataList))]


class _Match(C.Structure):
    _fields_ = [("item_type", C.c_uint8), ("value", C.c_char_p), ("start", C.c_size_t), ("end", C.c_size_t)]


class _Matches(C.Structure):
    _fields_ = [("items", C.POINTER(_Match)), ("count", C.c_size_t), ("_internal", C.c_void_p)]


class _ScanHit(C.Structure):
    # matchy_scan_hit_t (16 bytes): length in bits 0..23 of len_type, item type in bits 24..31
    _fields_ = [("start", C.c_uint32), ("len_type", C.c_uint32), ("value", C.c_uint32), ("kind", C.c_uint8),
                ("prefix_len", C.c_uint8), ("n_ids", C.c_uint16)]


class _ScanResult(C.Structure):
    _fields_ = [("hits", C.POINTER(_ScanHit)), ("n_hits", C.c_size_t), ("pattern_ids", C.POINTER(C.c_uint32)),
                ("data_offsets", C.POINTER(C.c_int64)), ("n_ids", C.c_size_t), ("lines", C.c_uint64),
                ("candidates", C.c_uint64), ("bytes", C.c_uint64), ("ip4_hits", C.POINTER(C.c_uint32 * 2)),
                ("n_ip4_hits", C.c_size_t), ("_internal", C.c_void_p)]


_lib = None


class _MultiBatch(C.Structure):
    _fields_ = [("seq", C.c_size_t), ("status", C.c_int32), ("result", _ScanResult), ("data", C.c_void_p), ("len", C.c_size_t),
                ("tag", C.c_void_p), ("payload", C.c_void_p), ("worker", C.c_size_t)]


class _MultiTotals(C.Structure):
    _fields_ = [("batches", C.c_uint64), ("bytes", C.c_uint64), ("lines", C.c_uint64), ("candidates", C.c_uint64), ("matches", C.c_uint64)]


_MULTI_ORDERED_FN = C.CFUNCTYPE(C.c_int32, C.c_void_p, C.POINTER(_MultiBatch))


def lib():
    """Load libmatchy_amd.so (built in-tree by matchy_amd.build). Raises if it is missing — no fallback."""
    global _lib
    if _lib is not None:
        return _lib
    # MATCHY_AMD_LIB: another build of the same library (kernel A/B experiments: tools/ab.sh); never a fallback
    path = Path(os.environ["MATCHY_AMD_LIB"]) if os.environ.get("MATCHY_AMD_LIB") else LIB_PATH
    if not path.exists():
        raise ImportError(f"{path} is missing: run `python -m matchy_amd.build` (hipcc, gfx950) first")
    L = C.CDLL(str(path))
    vp, cp, u8p = C.c_void_p, C.c_char_p, C.POINTER(C.c_uint8)
    sig = {
        "matchy_builder_new": (vp, []),
        "matchy_builder_set_case_insensitive": (C.c_int32, [vp, C.c_bool]),
        "matchy_builder_add": (C.c_int32, [vp, cp, cp]),
        "matchy_builder_set_description": (C.c_int32, [vp, cp]),
        "matchy_builder_save": (C.c_int32, [vp, cp]),
        "matchy_builder_build": (C.c_int32, [vp, C.POINTER(vp), C.POINTER(C.c_size_t)]),
        "matchy_builder_free": (None, [vp]),
        "matchy_builder_set_build_epoch": (C.c_int32, [vp, C.c_uint64]),
        "matchy_open": (vp, [cp]),
        "matchy_open_with_options": (vp, [cp, vp]),
        "matchy_open_buffer": (vp, [cp, C.c_size_t]),
        "matchy_close": (None, [vp]),
        "matchy_query": (_Result, [vp, cp]),
        "matchy_query_into": (None, [vp, cp, C.POINTER(_Result)]),
        "matchy_free_result": (None, [C.POINTER(_Result)]),
        "matchy_free_string": (None, [vp]),
        "matchy_result_to_json": (vp, [C.POINTER(_Result)]),
        "matchy_version": (cp, []),
        "matchy_format": (cp, [vp]),
        "matchy_has_ip_data": (C.c_bool, [vp]),
        "matchy_has_string_data": (C.c_bool, [vp]),
        "matchy_has_literal_data": (C.c_bool, [vp]),
        "matchy_has_glob_data": (C.c_bool, [vp]),
        "matchy_metadata": (vp, [vp]),
        "matchy_get_pattern_string": (vp, [vp, C.c_uint32]),
        "matchy_pattern_count": (C.c_size_t, [vp]),
        "matchy_extractor_create": (vp, [C.c_uint32]),
        "matchy_amd_extractor_create": (vp, [C.c_uint32, C.c_uint32]),
        "matchy_extractor_extract_chunk": (C.c_int32, [vp, cp, C.c_size_t, C.POINTER(_Matches)]),
        "matchy_matches_free": (None, [C.POINTER(_Matches)]),
        "matchy_extractor_free": (None, [vp]),
        "matchy_item_type_name": (cp, [C.c_uint8]),
        "matchy_scanner_create": (vp, [vp, C.c_uint32, C.c_int32]),
        "matchy_scanner_free": (None, [vp]),
        "matchy_scanner_scan": (C.c_int32, [vp, vp, C.c_size_t, C.POINTER(_ScanResult)]),
        "matchy_scanner_scan_device": (C.c_int32, [vp, vp, C.c_size_t, vp, C.c_uint32, C.POINTER(_ScanResult)]),
        "matchy_scanner_submit_device": (C.c_int32, [vp, vp, C.c_size_t, vp, C.c_uint32]),
        "matchy_scanner_wait": (C.c_int32, [vp, C.POINTER(_ScanResult)]),
        "matchy_scan_result_free": (None, [C.POINTER(_ScanResult)]),
        "matchy_scan_hit_to_json": (vp, [vp, C.POINTER(_ScanResult), C.c_size_t, cp, cp]),
        "matchy_scan_result_to_ndjson": (C.c_int32, [vp, C.POINTER(_ScanResult), cp, cp, C.POINTER(vp), C.POINTER(C.c_size_t)]),
        "matchy_scanner_set_profile": (None, [vp, C.c_bool]),
        "matchy_amd_ac_dfa_states": (C.c_int32, [vp]),
        "matchy_amd_suffix_filter": (C.c_int32, [vp]),
        "matchy_amd_pinned_alloc": (vp, [C.c_size_t]),
        "matchy_amd_pinned_free": (None, [vp]),
        "matchy_amd_host_register": (C.c_int32, [vp, C.c_size_t]),
        "matchy_amd_host_unregister": (None, [vp]),
        "matchy_scanner_set_slices": (None, [vp, C.c_int32]),
        "matchy_scanner_last_slices": (C.c_int32, [vp]),
        "matchy_scan_result_on_device": (C.c_bool, [C.POINTER(_ScanResult)]),
        "matchy_scanner_get_timing": (None, [vp, C.POINTER(C.c_float)]),
        "matchy_amd_last_error": (cp, []),
        "matchy_get_stats": (None, [vp, C.POINTER(_Stats)]),
        "matchy_clear_cache": (None, [vp]),
        "matchy_has_pattern_data": (C.c_bool, [vp]),
        "matchy_result_get_entry": (C.c_int32, [C.POINTER(_Result), C.POINTER(_Entry)]),
        "matchy_aget_value": (C.c_int32, [C.POINTER(_Entry), C.POINTER(_EntryData), C.POINTER(cp)]),
        "matchy_get_entry_data_list": (C.c_int32, [C.POINTER(_Entry), C.POINTER(C.POINTER(_EntryDataList))]),
        "matchy_free_entry_data_list": (None, [C.POINTER(_EntryDataList)]),
        "matchy_validate": (C.c_int32, [cp, C.c_int32, C.POINTER(vp)]),
        "matchy_builder_set_schema": (C.c_int32, [vp, cp]),
        "matchy_amd_device_count": (C.c_int32, []),
        "matchy_amd_device_numa_node": (C.c_int32, [C.c_int32]),
        "matchy_amd_bind_thread_to_device": (C.c_int32, [C.c_int32]),
        "matchy_amd_numa_cpus": (C.c_int32, [cp, cp, C.POINTER(C.c_int32), C.c_size_t]),
        "matchy_multi_scanner_create": (vp, [vp, C.c_uint32, C.POINTER(C.c_int32), C.c_size_t]),
        "matchy_multi_scanner_free": (None, [vp]),
        "matchy_multi_scanner_workers": (C.c_size_t, [vp]),
        "matchy_multi_scanner_worker_scanner": (vp, [vp, C.c_size_t]),
        "matchy_multi_scanner_submit": (C.c_int32, [vp, vp, C.c_size_t, vp, vp]),
        "matchy_multi_scanner_next": (C.c_int32, [vp, C.POINTER(_MultiBatch)]),
        "matchy_multi_scanner_submit_near": (C.c_int32, [vp, vp, C.c_size_t, vp, vp, C.c_int32]),
        "matchy_multi_scanner_worker_numa": (C.c_int32, [vp, C.c_size_t, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        "matchy_multi_scanner_pending": (C.c_size_t, [vp]),
        "matchy_multi_scanner_max_pending": (C.c_size_t, [vp]),
        "matchy_amd_unbind_thread": (C.c_int32, []),
        "matchy_multi_scanner_scan": (C.c_int32, [vp, vp, C.c_size_t, C.c_size_t, C.POINTER(_ScanResult)]),
        "matchy_multi_scanner_scan_file": (C.c_int32, [vp, cp, C.c_size_t, _MULTI_ORDERED_FN, vp, C.POINTER(_MultiTotals)]),
    }
    for name, (res, args) in sig.items():
        f = getattr(L, name)
        f.restype = res
        f.argtypes = args
    _lib = L
    return L


def last_error() -> str:
    return lib().matchy_amd_last_error().decode("utf-8", "replace")


def _take_string(ptr) -> str:
    if not ptr:
        return None
    s = C.string_at(ptr).decode("utf-8", "replace")
    lib().matchy_free_string(ptr)
    return s


class DatabaseBuilder:
    """Mirror of the reference builder behind matchy_builder_* (host-side, no GPU needed)."""

    def __init__(self, build_epoch=None, case_insensitive=False):
        self._h = lib().matchy_builder_new()
        if build_epoch is not None:
            lib().matchy_builder_set_build_epoch(self._h, build_epoch)
        if case_insensitive:
            lib().matchy_builder_set_case_insensitive(self._h, True)

    def add_entry(self, key: str, data: dict):
        rc = lib().matchy_builder_add(self._h, key.encode("utf-8"), json.dumps(data).encode("utf-8"))
        if rc != 0:
            raise ValueError(f"matchy_builder_add({key!r}) failed: rc={rc} {last_error()}")

    def set_description(self, text: str):
        lib().matchy_builder_set_description(self._h, text.encode("utf-8"))

    def build(self) -> bytes:
        buf = C.c_void_p()
        size = C.c_size_t()
        rc = lib().matchy_builder_build(self._h, C.byref(buf), C.byref(size))
        if rc != 0:
            raise ValueError(f"matchy_builder_build failed: rc={rc} {last_error()}")
        try:
            return C.string_at(buf, size.value)
        finally:
            C.CDLL(None).free(buf)

    def save(self, path: str):
        rc = lib().matchy_builder_save(self._h, str(path).encode())
        if rc != 0:
            raise OSError(f"matchy_builder_save failed: rc={rc} {last_error()}")

    def close(self):
        if self._h:
            lib().matchy_builder_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Database:
    """An opened .mxy database, uploaded once to device memory (matchy_open / matchy_open_buffer)."""

    def __init__(self, source):
        L = lib()
        if isinstance(source, (bytes, bytearray, memoryview)):
            b = bytes(source)
            self._h = L.matchy_open_buffer(b, len(b))
        else:
            self._h = L.matchy_open(str(source).encode())
        if not self._h:
            raise RuntimeError("matchy_open failed: " + last_error())

    def close(self):
        if self._h:
            lib().matchy_close(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @property
    def handle(self):
        return self._h

    def lookup(self, query: str):
        """Database::lookup → None | {'found': True, 'prefix_len': n, 'data': {...}} (matchy_query semantics)."""
        L = lib()
        r = L.matchy_query(self._h, query.encode("utf-8") if isinstance(query, str) else bytes(query))
        try:
            if not r.found:
                return None
            js = _take_string(L.matchy_result_to_json(C.byref(r)))
            return {"found": True, "prefix_len": r.prefix_len, "data": json.loads(js) if js else None}
        finally:
            L.matchy_free_result(C.byref(r))

    def query_json(self, query: str):
        """what `matchy query DB QUERY` prints (matchy_amd_query_json): (found, list) — one object per pattern that carries data
        (literal first, then globs by id), or one object for an IP hit with "cidr" and "prefix_len" added; [] when nothing matches"""
        L = lib()
        found = C.c_int32(0)
        L.matchy_amd_query_json.restype = C.c_void_p
        L.matchy_amd_query_json.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(C.c_int32)]
        p = L.matchy_amd_query_json(self._h, query.encode("utf-8") if isinstance(query, str) else bytes(query), C.byref(found))
        if not p:
            raise RuntimeError("matchy_amd_query_json failed: " + last_error())
        return bool(found.value), json.loads(_take_string(p))

    @staticmethod
    def _entry_value(ed):
        """matchy_entry_data_t -> (type, python value); maps / arrays yield their element count."""
        t = ed.type_
        v = ed.value
        if t == 2:
            return t, C.string_at(v.utf8_string, ed.data_size).decode("utf-8")
        if t == 4:
            return t, bytes(v.bytes[i] for i in range(ed.data_size))
        return t, {1: v.pointer, 3: v.double_value, 5: v.uint16, 6: v.uint32, 7: ed.data_size, 8: v.int32, 9: v.uint64,
                   10: int.from_bytes(bytes(v.uint128), "big"), 11: ed.data_size, 14: bool(v.boolean), 15: v.float_value}[t]

    def get_value(self, query: str, *path):
        """matchy_query + matchy_result_get_entry + matchy_aget_value -> (rc, type, value)."""
        L = lib()
        r = L.matchy_query(self._h, query.encode("utf-8"))
        try:
            e = _Entry()
            rc = L.matchy_result_get_entry(C.byref(r), C.byref(e))
            if rc != 0:
                return rc, None, None
            arr = (C.c_char_p * (len(path) + 1))(*[str(x).encode("utf-8") for x in path], None)
            ed = _EntryData()
            rc = L.matchy_aget_value(C.byref(e), C.byref(ed), arr)
            if rc != 0 or not ed.has_data:
                return rc, None, None
            t, v = self._entry_value(ed)
            return rc, t, v
        finally:
            L.matchy_free_result(C.byref(r))

    def entry_data_list(self, query: str):
        """matchy_get_entry_data_list flattened to [(type, value)], or None when the query has no result."""
        L = lib()
        r = L.matchy_query(self._h, query.encode("utf-8"))
        try:
            e = _Entry()
            if L.matchy_result_get_entry(C.byref(r), C.byref(e)) != 0:
                return None
            head = C.POINTER(_EntryDataList)()
            if L.matchy_get_entry_data_list(C.byref(e), C.byref(head)) != 0:
                return None
            out, node = [], head
            while node:
                out.append(self._entry_value(node.contents.entry_data))
                node = node.contents.next
            L.matchy_free_entry_data_list(head)
            return out
        finally:
            L.matchy_free_result(C.byref(r))

    def stats(self):
        st = _Stats()
        lib().matchy_get_stats(self._h, C.byref(st))
        return {n: getattr(st, n) for n, _ in _Stats._fields_}

    def has_ip_data(self):
        return bool(lib().matchy_has_ip_data(self._h))

    def has_literal_data(self):
        return bool(lib().matchy_has_literal_data(self._h))

    def has_glob_data(self):
        return bool(lib().matchy_has_glob_data(self._h))

    def format(self):
        return lib().matchy_format(self._h).decode()

    def metadata(self):
        return json.loads(_take_string(lib().matchy_metadata(self._h)))

    def pattern_count(self):
        return lib().matchy_pattern_count(self._h)

    def pattern_string(self, pid):
        return _take_string(lib().matchy_get_pattern_string(self._h, pid))


class Extractor:
    """Extractor::extract_from_chunk on the GPU (matchy_extractor_*). Returns (type_name, start, end, value)."""

    def __init__(self, flags=EXTRACT_ALL, min_domain_labels=2):
        # min_domain_labels: ExtractorBuilder::min_domain_labels (matchy-extractor/src/lib.rs:101-104), through the additive entry
        self._h = (lib().matchy_extractor_create(flags) if min_domain_labels == 2
                   else lib().matchy_amd_extractor_create(flags, min_domain_labels))
        if not self._h:
            raise RuntimeError("matchy_extractor_create failed: " + last_error())

    def extract_from_chunk(self, data: bytes):
        L = lib()
        m = _Matches()
        rc = L.matchy_extractor_extract_chunk(self._h, bytes(data), len(data), C.byref(m))
        if rc != 0:
            raise RuntimeError(f"matchy_extractor_extract_chunk failed: rc={rc} {last_error()}")
        try:
            return [(ITEM_TYPE_NAMES[m.items[i].item_type], m.items[i].start, m.items[i].end, m.items[i].value.decode("utf-8"))
                    for i in range(m.count)]
        finally:
            L.matchy_matches_free(C.byref(m))

    def close(self):
        if self._h:
            lib().matchy_extractor_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class ScanResult:
    def __init__(self, scanner, raw):
        self._scanner = scanner
        self._raw = raw
        self.lines = raw.lines
        self.candidates = raw.candidates
        self.bytes = raw.bytes
        self.n_ip4_hits = raw.n_ip4_hits   # fetch_mode 1 | 8: IPv4 results as compact records (not in _raw.hits)
        self.n_hits = raw.n_hits + raw.n_ip4_hits
        self._on_device = None

    @property
    def on_device(self):
        """fetch_mode 4: the arrays of the result are device pointers"""
        if self._on_device is None:
            self._on_device = bool(lib().matchy_scan_result_on_device(C.byref(self._raw))) if self._raw is not None else False
        return self._on_device

    def hits(self):
        """list of dict(start,end,type,kind,prefix_len,ip_data_offset,ids,offs) in canonical order."""
        r = self._raw
        out = []
        if self.on_device:
            raise RuntimeError("the hit records of this result are in device memory (fetch_mode 4): read them on the GPU")
        for i in range(r.n_ip4_hits if r.ip4_hits else 0):   # matchy_scan_ip4_hit_expand
            start, packed = r.ip4_hits[i]
            out.append(dict(start=start, end=start + ((packed >> 22) & 15) + 7, type=ITEM_TYPE_NAMES[2], kind="ip", prefix_len=packed >> 26,
                            ip_data_offset=packed & 0x3FFFFF, ids=[], offs=[]))
        if not r.hits:
            return out
        for i in range(r.n_hits):
            h = r.hits[i]
            ids = [r.pattern_ids[h.value + k] for k in range(h.n_ids)] if h.kind == 3 else []
            offs = [r.data_offsets[h.value + k] for k in range(h.n_ids)] if h.kind == 3 else []
            out.append(dict(start=h.start, end=h.start + (h.len_type & 0xFFFFFF), type=ITEM_TYPE_NAMES[h.len_type >> 24],
                            kind="ip" if h.kind == 2 else "pattern", prefix_len=h.prefix_len,
                            ip_data_offset=h.value if h.kind == 2 else 0, ids=ids, offs=offs))
        return out

    def ndjson(self, text: bytes, source="-"):
        L = lib()
        if self.on_device:
            raise RuntimeError("the hit records of this result are in device memory (fetch_mode 4): read them on the GPU")
        n = (self._raw.n_hits if self._raw.hits else 0) + (self._raw.n_ip4_hits if self._raw.ip4_hits else 0)
        if not self._raw.hits and self._raw.n_hits:
            return []   # counters only
        return [_take_string(L.matchy_scan_hit_to_json(self._scanner._h, C.byref(self._raw), i, text, source.encode())) for i in range(n)]

    def ndjson_text(self, text: bytes, source="-") -> bytes:
        """all matches as one NDJSON text in one call (matchy_scan_result_to_ndjson: what `matchy match` prints)"""
        L = lib()
        out, n = C.c_void_p(), C.c_size_t()
        rc = L.matchy_scan_result_to_ndjson(self._scanner._h, C.byref(self._raw), text, source.encode(), C.byref(out), C.byref(n))
        if rc != 0:
            raise RuntimeError(f"matchy_scan_result_to_ndjson failed ({rc}): " + last_error())
        try:
            return C.string_at(out.value, n.value)
        finally:
            L.matchy_free_string(out)

    def close(self):
        if self._raw is not None:
            if self._raw._internal:   # borrowed results (fetch_mode 0 / 1 / 9) own nothing: no call needed
                lib().matchy_scan_result_free(C.byref(self._raw))
            self._raw = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Scanner:
    """Bulk scan session (the device-side Worker). One per thread / stream."""

    def __init__(self, db: Database, extract_flags=0, device=0, profile=False):
        self._db = db
        self._h = lib().matchy_scanner_create(db.handle, extract_flags, device)
        if not self._h:
            raise RuntimeError("matchy_scanner_create failed: " + last_error())
        if profile:
            lib().matchy_scanner_set_profile(self._h, True)

    def scan(self, data: bytes) -> ScanResult:
        raw = _ScanResult()
        rc = lib().matchy_scanner_scan(self._h, bytes(data), len(data), C.byref(raw))
        if rc != 0:
            raise RuntimeError(f"matchy_scanner_scan failed: rc={rc} {last_error()}")
        return ScanResult(self, raw)

    def scan_ptr(self, host_ptr: int, nbytes: int) -> ScanResult:
        """Like scan() for a host buffer given by address (e.g. numpy / torch CPU storage) — no Python-side copy."""
        raw = _ScanResult()
        rc = lib().matchy_scanner_scan(self._h, host_ptr, nbytes, C.byref(raw))
        if rc != 0:
            raise RuntimeError(f"matchy_scanner_scan failed: rc={rc} {last_error()}")
        return ScanResult(self, raw)

    def scan_device(self, device_ptr: int, nbytes: int, stream: int = 0, fetch_mode=1) -> ScanResult:
        """fetch_mode: 0 counters only, 1 hit records in device order (borrowed from the scanner's pinned buffers), 3 hit records
        in canonical order (owned copy), 1 | 8 like 1 with the IPv4 results as 8-byte records (n_ip4_hits; hits() / ndjson() read
        both arrays), 4 the records stay in device memory (result.on_device: `_raw.hits` / `pattern_ids` /
        `data_offsets` are device addresses; hits() raises)."""
        raw = _ScanResult()
        rc = lib().matchy_scanner_scan_device(self._h, device_ptr, nbytes, stream, fetch_mode, C.byref(raw))
        if rc != 0:
            raise RuntimeError(f"matchy_scanner_scan_device failed: rc={rc} {last_error()}")
        return ScanResult(self, raw)

    def submit_device(self, device_ptr: int, nbytes: int, stream: int = 0, fetch_mode=1):
        rc = lib().matchy_scanner_submit_device(self._h, device_ptr, nbytes, stream, fetch_mode)
        if rc != 0:
            raise RuntimeError(f"matchy_scanner_submit_device failed: rc={rc} {last_error()}")

    def wait(self) -> ScanResult:
        raw = _ScanResult()
        rc = lib().matchy_scanner_wait(self._h, C.byref(raw))
        if rc != 0:
            raise RuntimeError(f"matchy_scanner_wait failed: rc={rc} {last_error()}")
        return ScanResult(self, raw)

    def set_slices(self, n: int):
        """scan_device cuts large batches into slices (tail of one slice beside the streaming pass of the next): 0 = default, 1 = never, n = n equal slices."""
        lib().matchy_scanner_set_slices(self._h, n)

    def last_slices(self) -> int:
        return lib().matchy_scanner_last_slices(self._h)

    def timing_ms(self):
        out = (C.c_float * 5)()
        lib().matchy_scanner_get_timing(self._h, out)
        return dict(anchor=out[0], validate=out[1], rare=out[2], lookup=out[3], total=out[4])

    def close(self):
        if self._h:
            lib().matchy_scanner_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _WorkerScanner:
    """the scanner of one worker of a MultiScanner, as far as ScanResult needs it (matchy_scan_hit_to_json)"""

    def __init__(self, h):
        self._h = h


class MultiScanner:
    """One scan session over several devices (matchy_multi_scanner_*): the reader -> per-device workers -> ordered gather of the
    reference's process_files_parallel (processing/parallel.rs:494-505) behind the C ABI. `devices` may repeat an ordinal (several
    batches of one GPU in flight)."""

    def __init__(self, db: Database, devices=(0,), extract_flags=0):
        self._db = db
        arr = (C.c_int32 * len(devices))(*devices)
        self._h = lib().matchy_multi_scanner_create(db.handle, extract_flags, arr, len(devices))
        if not self._h:
            raise RuntimeError("matchy_multi_scanner_create failed: " + last_error())
        self.workers = lib().matchy_multi_scanner_workers(self._h)

    def scan(self, data: bytes, batch_bytes: int = 0) -> ScanResult:
        """one buffer through all workers; the merged result has offsets into `data` (= Scanner.scan of the same bytes)"""
        raw = _ScanResult()
        rc = lib().matchy_multi_scanner_scan(self._h, bytes(data), len(data), batch_bytes, C.byref(raw))
        if rc != 0:
            raise RuntimeError(f"matchy_multi_scanner_scan failed ({rc}): " + last_error())
        return ScanResult(_WorkerScanner(lib().matchy_multi_scanner_worker_scanner(self._h, 0)), raw)

    def submit_ptr(self, host_ptr: int, nbytes: int, tag: int = 0, numa_node: int = -1):
        """queue one newline-aligned batch that lives at a host address (e.g. a pinned torch tensor); take it back with next().
        Blocks while max_pending() batches are out: a caller that submits and gathers on one thread calls next() when
        pending() has reached max_pending(). numa_node: the node the bytes live on (-1 = anywhere)."""
        rc = lib().matchy_multi_scanner_submit_near(self._h, host_ptr, nbytes, tag, None, numa_node)
        if rc != 0:
            raise RuntimeError(f"matchy_multi_scanner_submit failed ({rc}): " + last_error())

    def pending(self) -> int:
        return lib().matchy_multi_scanner_pending(self._h)

    def max_pending(self) -> int:
        return lib().matchy_multi_scanner_max_pending(self._h)

    def worker_numa(self):
        """[(NUMA node of the worker's GPU, CPUs its thread is bound to)] per worker"""
        out = []
        for w in range(self.workers):
            node, cpus = C.c_int32(-1), C.c_int32(0)
            lib().matchy_multi_scanner_worker_numa(self._h, w, C.byref(node), C.byref(cpus))
            out.append((node.value, cpus.value))
        return out

    def next(self, want_hits=False):
        """the next batch in submission order: dict(seq, tag, worker, lines, candidates, n_hits[, hits]) or None when nothing is pending"""
        b = _MultiBatch()
        rc = lib().matchy_multi_scanner_next(self._h, C.byref(b))
        if rc == 0:
            return None
        if rc != 1 or b.status != 0:
            raise RuntimeError(f"matchy_multi_scanner_next failed ({rc}, batch status {b.status}): " + last_error())
        out = dict(seq=b.seq, tag=int(b.tag or 0), worker=b.worker, lines=int(b.result.lines), candidates=int(b.result.candidates),
                   n_hits=int(b.result.n_hits + b.result.n_ip4_hits), nbytes=int(b.len))
        r = ScanResult(_WorkerScanner(lib().matchy_multi_scanner_worker_scanner(self._h, b.worker)), b.result)
        if want_hits:
            out["hits"] = r.hits()
        r.close()
        return out

    def scan_file(self, path: str, batch_bytes: int = 0, on_batch=None):
        """scan one file (or "-"); on_batch(offset, nbytes, hits, lines, candidates) is called per batch in file order. Returns the totals."""
        L = lib()
        me = self

        def cb(_user, bp):
            b = bp.contents
            if on_batch is not None:
                r = ScanResult(_WorkerScanner(L.matchy_multi_scanner_worker_scanner(me._h, b.worker)), b.result)
                try:
                    on_batch(int(b.tag or 0), int(b.len), r.hits(), int(b.result.lines), int(b.result.candidates))
                finally:
                    r._raw = None   # the library releases the result when the callback returns
            return 0

        fn = _MULTI_ORDERED_FN(cb)
        tot = _MultiTotals()
        rc = L.matchy_multi_scanner_scan_file(self._h, os.fsencode(path), batch_bytes, fn, None, C.byref(tot))
        if rc != 0:
            raise RuntimeError(f"matchy_multi_scanner_scan_file failed ({rc}): " + last_error())
        return dict(batches=tot.batches, bytes=tot.bytes, lines=tot.lines, candidates=tot.candidates, matches=tot.matches)

    def close(self):
        if self._h:
            lib().matchy_multi_scanner_free(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def numa_cpus(sysfs_root: str, pci_bus_id: str):
    """CPUs on the NUMA node of a PCI device under any sysfs root (matchy_amd_numa_cpus: the mapping the workers bind with)"""
    out = (C.c_int32 * 4096)()
    n = lib().matchy_amd_numa_cpus(os.fsencode(sysfs_root), pci_bus_id.encode(), out, 4096)
    return [out[i] for i in range(max(0, min(n, 4096)))]
