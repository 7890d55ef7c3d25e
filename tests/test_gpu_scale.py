"""GPU tests at scale (-m gpu): dense / pathological inputs that overflow the work lists and force the regrow-and-rescan
path, and the BASELINE configs[1] workload at a size that exercises the > 1 GiB host path, checked against the oracle
and through size-independent properties (idempotence, sharding invariance, additivity of the line count)."""
import os
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def M():
    import matchy_amd
    matchy_amd.lib()
    return matchy_amd


# (the reference's domain pass is quadratic in the length of a dot-rich run, and so is the oracle: keep such runs short)
DENSE = [
    (b"1.1.1.1 ", 300_000), (b",1.1.,1.1.", 200_000), (b"10.20.30.40,", 150_000), (b"1.2.3.4.5 ", 100_000), (b"999.999.999.999 ", 60_000),
    (b"a.com ", 300_000), (b"a.b.c.d.e.com/", 100_000), (b"x.html y.css z.js ", 100_000), (b"www.example.co.uk\n", 80_000),
    (b"u@a.co ", 200_000), (b"@", 500_000), (b"::1 ", 300_000), (b" 2001:db8::1", 120_000), (b":", 700_000), (b".", 1_500), (b"a.", 1_000),
    (b"5d41402abc4b2a76b9719d911017c592 ", 40_000), (b"a" * 26 + b" ", 50_000), (b"1" * 34 + b" ", 3_000), (b"0x" + b"ab" * 20 + b" ", 30_000),
    (b"\n", 1_000_000), (b"x", 1_500_000), (b" ", 1_500_000),
]


def test_dense_pathological_inputs(M, oracle):
    ex = M.Extractor()
    for unit, reps in DENSE:
        buf = unit * reps
        got = ex.extract_from_chunk(buf)
        want = oracle.extract(buf)
        assert len(got) == len(want), (unit, len(got), len(want))
        assert got == want, unit
    ex.close()


def test_raw_window_edges(M, oracle):
    # candidates that straddle the 4 KiB wrap of the streaming kernel's LDS window and its 1 KiB blocks, at every offset
    ex = M.Extractor()
    payloads = [b"10.20.30.40", b"255.255.255.255", b"evil.example.com", b"a.io", b"2001:db8::8a2e:370:7334"]
    for edge in (1024, 4096, 8192, 12288, 16384, 20480):
        for pl in payloads:
            for shift in range(0, len(pl) + 18):
                pre = edge - shift + 8
                buf = b"q" * (pre - 1) + b" " + pl + b" " + b"1.2.3.4 b.org " * 40
                assert ex.extract_from_chunk(buf) == oracle.extract(buf), (edge, pl, shift)
    ex.close()


def _rebased(hits, base):
    out = []
    for h in hits:
        h = dict(h)
        h["start"] += base
        h["end"] += base
        out.append(h)
    return out


def test_full_size_parity_and_properties(M, oracle):
    """6 M lines of the C2 log (> 1 GiB: scan_host cuts it into newline-aligned pieces) against the oracle, plus
    idempotence and sharding invariance of the hit set."""
    from matchy_amd import sharding
    from tools import synth
    cfg = synth.config("c2")
    blob = synth.build_db(cfg)
    lines = int(os.environ.get("MXY_SCALE_LINES", "6000000"))
    log = synth.make_log(cfg, 0, lines)
    assert lines < 6_000_000 or len(log) > (1 << 30)
    db = M.Database(blob)
    sc = M.Scanner(db)
    res = sc.scan(log)
    hits = res.hits()
    stats = (res.lines, res.candidates, res.bytes)
    res.close()
    assert stats[0] == lines and stats[2] == len(log)
    # oracle on the same bytes (all host cores, no cache: plain reference semantics)
    odb = oracle.Database(blob)
    want, _, st = odb.scan(log, threads=min(len(os.sched_getaffinity(0)), 16), cache=0, want_json=False)
    assert (st.lines, st.candidates) == stats[:2]
    assert len(hits) == len(want)
    assert hits == want
    # idempotence
    res2 = sc.scan(log)
    assert res2.hits() == hits
    res2.close()
    # sharding invariance (N4): three newline-aligned ranges, rebased and concatenated
    parts = sharding.split_at_newlines(log, 3)
    merged, nl = [], 0
    for a, b in parts:
        r = sc.scan(log[a:b])
        merged += _rebased(r.hits(), a)
        nl += r.lines
        r.close()
    assert nl == lines
    assert merged == hits
    sc.close(); db.close()


@pytest.mark.parametrize("size", ["headline", "largest"])
def test_headline_batch_as_one_launch(M, oracle, size):
    """BASELINE configs[1] exactly as bench.py runs it: 10 M lines (1.8 GB) resident in HBM, ONE launch of every kernel
    (matchy_scanner_scan_device; the host-buffer entry would cut the log into < 1 GiB pieces), and the largest batch one
    launch accepts (just below 2^31 - 2^16 bytes: 32-bit positions). Counters and the whole hit set against the oracle on
    the same bytes; then the property the sharding relies on: the two halves of the batch, cut at a newline and scanned
    as launches of their own, give the same counters and hits."""
    import ctypes
    from tools import synth
    cfg = synth.config("c2")
    blob = synth.build_db(cfg)
    lines = int(os.environ.get("MXY_HEADLINE_LINES", "10000000"))
    if size == "largest":
        log = synth.make_log(cfg, 0, lines + lines // 5)
        if lines >= 10_000_000:
            assert len(log) >= 0x7FFF0000
        log = log[: log.rfind(b"\n", 0, min(len(log), 0x7FFF0000 - 1)) + 1]
    else:
        log = synth.make_log(cfg, 0, lines)
    assert lines < 10_000_000 or len(log) > (3 << 29)
    want, _, st = oracle.Database(blob).scan(log, threads=min(len(os.sched_getaffinity(0)), 16), cache=0, want_json=False)
    db = M.Database(blob)
    sc = M.Scanner(db)
    hip = ctypes.CDLL("libamdhip64.so")
    dptr = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dptr), ctypes.c_size_t(len(log) + 64)) == 0
    assert hip.hipMemcpy(dptr, log, ctypes.c_size_t(len(log)), 1) == 0
    r = sc.scan_device(dptr.value, len(log), fetch_mode=3)
    got = r.hits()
    assert (r.lines, r.candidates, r.n_hits) == (st.lines, st.candidates, len(want))
    r.close()
    assert got == want
    assert sc.last_slices() == 1   # the default: every kernel launched once over the whole batch
    # cut into three and eight slices (tails beside the next slice's k_anchor): the same records
    for ns in (3, 8):
        sc.set_slices(ns)
        r = sc.scan_device(dptr.value, len(log), fetch_mode=1)
        assert sc.last_slices() == ns
        assert (r.lines, r.candidates, r.n_hits) == (st.lines, st.candidates, len(want))
        key = lambda h: (h["start"], h["end"], h["type"])
        assert sorted(r.hits(), key=key) == sorted(want, key=key)
        r.close()
    sc.set_slices(0)
    # halves: 16-byte aligned start of the second one (the entry requires it), so the cut is moved to a newline that
    # is followed by an aligned offset by scanning the second half from a copy
    cut = log.rfind(b"\n", 0, len(log) // 2) + 1
    r1 = sc.scan_device(dptr.value, cut, fetch_mode=3)
    h1, c1 = r1.hits(), (r1.lines, r1.candidates)
    r1.close()
    d2 = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(d2), ctypes.c_size_t(len(log) - cut + 64)) == 0
    assert hip.hipMemcpy(d2, ctypes.c_void_p(dptr.value + cut), ctypes.c_size_t(len(log) - cut), 3) == 0
    r2 = sc.scan_device(d2.value, len(log) - cut, fetch_mode=3)
    h2, c2 = r2.hits(), (r2.lines, r2.candidates)
    r2.close()
    assert (c1[0] + c2[0], c1[1] + c2[1]) == (st.lines, st.candidates)
    assert h1 + _rebased(h2, cut) == want
    # The whole batch again, twice, through the same scanner: from its second launch on k_anchor starts with what the scanner learnt from the
    # previous one — every wave owns its first chunk of the domain list without a reservation (TokParams::dom_static, the counter preset by
    # k_finish or by the scan), the sparse lists start with chunks sized from the previous batch. Then a batch without a single domain (the
    # waves mark their own chunks unused; the scanner drops the preset), then the log once more: the same counters and records every time.
    if size == "headline":
        for rnd in range(2):
            r = sc.scan_device(dptr.value, len(log), fetch_mode=3)
            assert (r.lines, r.candidates, r.n_hits) == (st.lines, st.candidates, len(want)), rnd
            if rnd == 1:
                assert r.hits() == want
            r.close()
        plain = (b"the quick brown fox jumps over the lazy dog " * 4 + b"\n") * 1_000_000
        d3 = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(d3), ctypes.c_size_t(len(plain) + 64)) == 0
        assert hip.hipMemcpy(d3, plain, ctypes.c_size_t(len(plain)), 1) == 0
        for _ in range(2):
            r = sc.scan_device(d3.value, len(plain), fetch_mode=3)
            assert (r.lines, r.n_hits) == (1_000_000, 0)
            r.close()
        hip.hipFree(d3)
        r = sc.scan_device(dptr.value, len(log), fetch_mode=3)
        assert (r.lines, r.candidates, r.n_hits) == (st.lines, st.candidates, len(want)) and r.hits() == want
        r.close()
    sc.close(); db.close()
    hip.hipFree(dptr); hip.hipFree(d2)


def test_single_stream_fallback_gives_the_same_records(M):
    """MATCHY_AMD_NO_FORK=1 keeps the device-resident entry on the caller's stream (the scanner's two extra streams unused). The
    switch is read once per process, so the single-stream scan runs in a child process (one at a time: the GPU box allows few
    processes on the card) and hands back a digest of its records, as does a second child with the fork (test_scan_device_fetch_modes and the parity tests
    hold the forked entry against the oracle)."""
    import subprocess
    import sys
    code = r'''
import ctypes, hashlib, sys
sys.path.insert(0, ".")
import matchy_amd as M
from tools import synth
cfg = synth.config(sys.argv[1])
db = M.Database(synth.build_db(cfg))
log = synth.make_log(cfg, 0, 60000)
sc = M.Scanner(db)
hip = ctypes.CDLL("libamdhip64.so")
d = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(len(log) + 64)) == 0
assert hip.hipMemcpy(d, log, ctypes.c_size_t(len(log)), 1) == 0
r = sc.scan_device(d.value, len(log), fetch_mode=3)
print(hashlib.sha256(repr((r.lines, r.candidates, r.hits())).encode()).hexdigest())
'''
    root = Path(__file__).resolve().parent.parent
    for cfgname in ("c2/10", "c4/10"):
        digests = []
        for env_extra in ({}, {"MATCHY_AMD_NO_FORK": "1"}):
            env = {k: v for k, v in os.environ.items() if k != "MATCHY_AMD_NO_FORK"}
            env.update(env_extra)
            out = subprocess.run([sys.executable, "-c", code, cfgname], cwd=root, env=env, capture_output=True, text=True, timeout=600)
            assert out.returncode == 0, out.stderr[-2000:]
            digests.append(out.stdout.strip().splitlines()[-1])
        assert digests[0] == digests[1]


@pytest.mark.parametrize("cfgname,mirror", [("c4/10", "64"), ("c2/10", "64"), ("c2/10", "0")])
def test_scan_device_fetch_modes(M, oracle, cfgname, mirror, monkeypatch):
    """matchy_scanner_scan_device on a buffer that lives in HBM: fetch_mode 0 (counts), 1 (records straight from the
    pinned mirror that k_pack fills, device order) and 3 (GPU-sorted copy) agree with each other and with the oracle;
    a mirror that is too small (64 records) makes mode 1 take the copy path and grow the mirror for the next scan."""
    import ctypes
    from tools import synth
    if mirror != "0":
        monkeypatch.setenv("MATCHY_AMD_MIRROR_RECS", mirror)   # read once per process: only the first parametrisation
    cfg = synth.config(cfgname)
    blob = synth.build_db(cfg)
    log = synth.make_log(cfg, 0, 60000)
    want, _, st = oracle.Database(blob).scan(log, want_json=False)
    db = M.Database(blob)
    sc = M.Scanner(db)
    # device buffer through the HIP runtime the library itself uses (no torch in this process)
    hip = ctypes.CDLL("libamdhip64.so")
    dptr = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dptr), ctypes.c_size_t(len(log) + 64)) == 0
    assert hip.hipMemcpy(dptr, log, ctypes.c_size_t(len(log)), 1) == 0

    class _D:
        def data_ptr(self):
            return dptr.value
    d = _D()
    key = lambda h: (h["start"], h["end"], h["type"])
    for rep in range(3):  # repeated: borrowed buffers are reused, the mirror may grow between scans
        r0 = sc.scan_device(d.data_ptr(), len(log), fetch_mode=0)
        assert (r0.lines, r0.candidates, r0.n_hits) == (st.lines, st.candidates, len(want))
        r0.close()
        r1 = sc.scan_device(d.data_ptr(), len(log), fetch_mode=1)
        h1 = r1.hits()
        r1.close()
        r3 = sc.scan_device(d.data_ptr(), len(log), fetch_mode=3)
        h3 = r3.hits()
        r3.close()
        assert h3 == want
        assert sorted(h1, key=key) == sorted(want, key=key)
        # fetch_mode 1 | 8: the IPv4 results as compact 8-byte records in an array (and a pinned mirror) of their own — with the
        # 64-record mirror the first scan takes the copy path and grows it
        r9 = sc.scan_device(d.data_ptr(), len(log), fetch_mode=9)
        assert (r9.lines, r9.candidates, r9.n_hits) == (st.lines, st.candidates, len(want))
        assert r9.n_ip4_hits == sum(1 for h in want if h["type"] == "IPv4") > 0
        assert sorted(r9.hits(), key=key) == sorted(want, key=key)
        r9.close()
        # fetch_mode 4: the records stay in device memory (borrowed device pointers); copied back here they are the same set
        r4 = sc.scan_device(d.data_ptr(), len(log), fetch_mode=4)
        assert (r4.lines, r4.candidates, r4.n_hits) == (st.lines, st.candidates, len(want))
        buf = (M._ScanHit * r4.n_hits)()
        src = ctypes.cast(r4._raw.hits, ctypes.c_void_p)
        assert hip.hipMemcpy(buf, src, ctypes.c_size_t(16 * r4.n_hits), 2) == 0
        recs4 = sorted((b.start, b.start + (b.len_type & 0xFFFFFF), M.ITEM_TYPE_NAMES[b.len_type >> 24], b.kind, b.prefix_len, b.n_ids) for b in buf)
        # device pointers: the host-side accessors refuse them
        assert r4.on_device
        with pytest.raises(RuntimeError):
            r4.hits()
        with pytest.raises(RuntimeError):
            r4.ndjson(log)
        r4.close()
        # sliced scans through every fetch mode (60 000 lines = 11 MB: explicit slice counts are honoured down to 8 KiB slices)
        for ns in (3, 8):
            sc.set_slices(ns)
            r1 = sc.scan_device(d.data_ptr(), len(log), fetch_mode=1)
            assert sc.last_slices() == ns
            assert sorted(r1.hits(), key=key) == sorted(want, key=key)
            r1.close()
            r0 = sc.scan_device(d.data_ptr(), len(log), fetch_mode=0)
            assert (r0.lines, r0.candidates, r0.n_hits) == (st.lines, st.candidates, len(want))
            r0.close()
        sc.set_slices(0)
        assert recs4 == sorted((h["start"], h["end"], h["type"], 2 if h["kind"] == "ip" else 3, h["prefix_len"], len(h["ids"])) for h in want)
    sc.close(); db.close()
    hip.hipFree(dptr)


def test_submit_wait_overlapping_batches(M, oracle):
    """matchy_scanner_submit_device / matchy_scanner_wait: three scanners, each on its own stream, with different batches in
    flight at the same time; every batch's result equals the oracle's (and the synchronous entry's)."""
    import ctypes
    from tools import synth
    cfg = synth.config("c4/10")
    blob = synth.build_db(cfg)
    logs = [synth.make_log(cfg, a, 40000) for a in (0, 40000, 80000)]
    odb = oracle.Database(blob)
    wants = [odb.scan(lg, want_json=False) for lg in logs]
    db = M.Database(blob)
    hip = ctypes.CDLL("libamdhip64.so")
    scs, ptrs, streams = [], [], []
    for lg in logs:
        scs.append(M.Scanner(db))
        p = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(p), ctypes.c_size_t(len(lg) + 64)) == 0
        assert hip.hipMemcpy(p, lg, ctypes.c_size_t(len(lg)), 1) == 0
        ptrs.append(p)
        st = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0   # hipStreamNonBlocking
        streams.append(st)
    for rep in range(3):
        for sc, p, st, lg in zip(scs, ptrs, streams, logs):
            sc.submit_device(p.value, len(lg), stream=st.value, fetch_mode=3 if rep else 1)
        for sc, (want, _, stats) in zip(scs, wants):
            r = sc.wait()
            hits = r.hits()
            assert (r.lines, r.candidates, r.n_hits) == (stats.lines, stats.candidates, len(want))
            r.close()
            key = lambda h: (h["start"], h["end"], h["type"])
            assert sorted(hits, key=key) == sorted(want, key=key)
            if rep:
                assert hits == want
    with pytest.raises(RuntimeError):
        scs[0].wait()   # nothing submitted
    for sc in scs:
        sc.close()
    for p, st in zip(ptrs, streams):
        hip.hipStreamDestroy(st)
        hip.hipFree(p)
    db.close()


def test_config5_full_database(M, oracle):
    """BASELINE configs[4] at its real database size: 10 M indicators (9 M CIDR-heavy IPs + 1 M domains), the record width
    chosen by the node count of the finished tree (no environment hook), 1 M log lines against the oracle: counters and
    the complete hit set. Nearly every line hits (the /24 bitmap filters little): the dense output path."""
    from tools import synth
    cfg = synth.config("c5")
    blob = synth.build_db(cfg)
    lines = int(os.environ.get("MXY_C5_LINES", "1000000"))
    log = synth.make_log(cfg, 0, lines)
    db = M.Database(blob)
    md = db.metadata()
    assert md["node_count"] > (1 << 24) and md["record_size"] == 28, (md["node_count"], md["record_size"])
    sc = M.Scanner(db)
    res = sc.scan(log)
    hits = res.hits()
    stats = (res.lines, res.candidates)
    res.close()
    odb = oracle.Database(blob)
    want, _, st = odb.scan(log, threads=min(len(os.sched_getaffinity(0)), 16), cache=0, want_json=False)
    assert stats == (st.lines, st.candidates)
    assert len(hits) == len(want) and len(want) > lines
    assert hits == want
    # the same log resident in HBM with compact IPv4 records (fetch_mode 1 | 8): more records than the first pinned mirror holds
    # (copy path, then a larger mirror), then straight from the mirror the lookup kernel writes
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    dptr = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dptr), ctypes.c_size_t(len(log) + 64)) == 0
    assert hip.hipMemcpy(dptr, log, ctypes.c_size_t(len(log)), 1) == 0
    key = lambda h: (h["start"], h["end"], h["type"], h["prefix_len"], h["ip_data_offset"])
    want_sorted = sorted(want, key=key)
    for rep in range(2):
        r9 = sc.scan_device(dptr.value, len(log), fetch_mode=9)
        assert (r9.lines, r9.candidates, r9.n_hits) == (st.lines, st.candidates, len(want))
        assert r9.n_ip4_hits == sum(1 for h in want if h["type"] == "IPv4") > lines // 2
        assert sorted(r9.hits(), key=key) == want_sorted
        r9.close()
    hip.hipFree(dptr)
    sc.close(); db.close()


@pytest.mark.parametrize("cfgname", ["c3", "c3b", "c4"])
def test_full_database_multi_batch(M, oracle, cfgname):
    """BASELINE configs[2] (1 M domain / hash literals; c3b: the same keys written `glob:` = 1 M Aho-Corasick literals with substring
    semantics) and configs[3] (100 K indicators incl. 10 K globs) at full database size. Three consecutive batches of the log through
    one scanner's host-buffer entry (buffers and lists are reused between batches), each against the oracle; then the first batch
    (300 K lines) and a HASH-DENSE batch (two or three file hashes per line: more than 256 K long-token anchors, the point where the
    lean lookup pass keeps the automaton walk instead of handing unwalked candidates to the glob pass) through every device-resident
    entry — forked with the early glob pass and the undecided domains' chain on their own streams, sliced, submitted, compact records —
    against the same oracle results."""
    from tools import synth
    from tests.test_gpu_parity import _device_entries
    cfg = synth.config(cfgname)
    blob = synth.build_db(cfg)
    db = M.Database(blob)
    sc = M.Scanner(db)
    odb = oracle.Database(blob)
    threads = min(len(os.sched_getaffinity(0)), 16)
    per = 300000
    total = 0
    first = None
    for bi in range(3):
        log = synth.make_log(cfg, bi * per, per if bi != 1 else per // 3)   # a short batch in the middle
        res = sc.scan(log)
        hits = res.hits()
        stats = (res.lines, res.candidates)
        res.close()
        want, _, st = odb.scan(log, threads=threads, cache=0, want_json=False)
        assert stats == (st.lines, st.candidates)
        assert hits == want
        total += len(want)
        if bi == 0:
            first = (log, want, stats)
    assert total > 1000
    _device_entries(sc, first[0], first[1], None, first[2], slices=(3,))
    dense = synth.make_log(cfg, 7_000_000, 300000, shape="hash-dense")
    want, _, st = odb.scan(dense, threads=threads, cache=0, want_json=False)
    assert st.candidates > 600000 and len(want) > 1000
    res = sc.scan(dense)
    assert (res.lines, res.candidates) == (st.lines, st.candidates)
    assert res.hits() == want
    res.close()
    _device_entries(sc, dense, want, None, (st.lines, st.candidates), slices=(3,))
    sc.close(); db.close()


def test_hostile_long_runs(M, oracle):
    """Single tokens of megabytes (one lane used to walk them alone, a dependent load per 8 bytes: 76-88 ms each): the wave
    walks the bulk of such a run together (coop_domain_skip / coop_email_skip). Same items as the oracle, and fast."""
    import time
    cases = [
        ("1 MB label", (b"a" * 1_000_000 + b".com ") * 4),
        ("2 MB of short labels", b"x " + b"a." * 1_000_000 + b"com "),
        ("2 MB local part", b"x" * 2_000_000 + b"@a.com "),
        ("long run with a bad label deep inside", b"b" * 700_000 + b"..c" + b"d" * 700_000 + b".org "),
        ("long run ending in a dash label", b"-" + b"e" * 900_000 + b".net y"),
        ("long last label after a dot", b"x.c" + b"d" * 700_000 + b".org " + b"y.c" + b"d" * 700_000 + b" "),
        ("long local part with two dots", b"y" * 600_000 + b".." + b"z" * 600_000 + b"@b.org "),
        ("digits only local part", b"1" * 900_000 + b"@c.net "),
        ("non-ASCII inside a long run", b" " + b"f" * 500_000 + "é".encode() + b"g" * 500_000 + b".com "),
        ("run that starts at the buffer start", b"h" * 300_000 + b".io"),
        ("overlong sequence deep inside", b" " + b"i" * 400_000 + b"\xe0\x80\x80" + b"j" * 400_000 + b".com "),
        ("surrogate deep inside", b" " + b"k" * 400_001 + b"\xed\xa0\x80" + b"l" * 400_000 + b".com "),
        ("truncated sequence at a step edge", b" " + b"m" * 63 + b"\xf0\x9f\x98" + b"n" * 300_000 + b".com "),
        ("four-byte characters across step edges", b" " + ("o" * 61 + "😀").encode() * 4000 + b".com "),
        ("stray continuation byte", b" " + b"p" * 200_000 + b"\x80" + b"q" * 200_000 + b".com "),
    ]
    ex = M.Extractor()
    ex.extract_from_chunk(b"warm.up.example.com 1.2.3.4 a@b.com")
    for name, buf in cases:
        t0 = time.perf_counter()
        got = ex.extract_from_chunk(buf)
        dt = time.perf_counter() - t0
        want = oracle.extract(buf)
        assert [(t, s, e) for t, s, e, v in got] == [(t, s, e) for t, s, e, v in want], name
        assert [v for *_, v in got] == [v for *_, v in want], name
        assert dt < 0.05, (name, dt)   # includes the upload of the buffer; the old single-lane walk took 0.08 s and more
    ex.close()


@pytest.mark.parametrize("env", [{"MATCHY_AMD_L24": "0"}, {"MATCHY_AMD_LEAF_MB": "0"}, {"MATCHY_AMD_LEAF_MB": "1"}, {"MATCHY_AMD_L24": "1"}])
def test_ipv4_lookup_table_variants(M, oracle, monkeypatch, env):
    """The IPv4 lookup tables are chosen at open: 16-level table + walk (small trees, MATCHY_AMD_L24=0), /24 table + walk (no leaf
    tables: MATCHY_AMD_LEAF_MB=0), /24 table with leaf tables for SOME /24s and the walk for the rest (a cap of 1 MB = 512 leaf tables),
    everything (default / forced). All of them against the oracle on the CIDR-heavy mix (dense candidates -> k_lookup_ip) and on the
    headline mix (sparse candidates -> looked up inside k_anchor), scans through both entries and single queries."""
    from tools import synth
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    for cfgname, lines in (("c5/100", 20000), ("c2/5", 20000)):
        cfg = synth.config(cfgname)
        blob = synth.build_db(cfg)
        log = synth.make_log(cfg, 0, lines)
        want, _, st = oracle.Database(blob).scan(log, want_json=False)
        db = M.Database(blob)
        sc = M.Scanner(db)
        r = sc.scan(log)
        assert (r.lines, r.candidates) == (st.lines, st.candidates)
        assert r.hits() == want
        r.close()
        import ctypes
        hip = ctypes.CDLL("libamdhip64.so")
        d = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(len(log) + 64)) == 0
        assert hip.hipMemcpy(d, log, ctypes.c_size_t(len(log)), 1) == 0
        r = sc.scan_device(d.value, len(log), fetch_mode=3)
        assert (r.lines, r.candidates) == (st.lines, st.candidates)
        assert r.hits() == want
        r.close()
        hip.hipFree(d)
        odb = oracle.Database(blob)
        seen = 0
        for h in want[:400]:
            if h["kind"] == "ip":
                q = log[h["start"]:h["end"]].decode()
                w, g = odb.lookup(q), db.lookup(q)
                assert g == {"found": True, "prefix_len": w["prefix_len"], "data": w["data"]}, q
                seen += 1
        assert seen > 20
        sc.close(); db.close()


@pytest.mark.parametrize("shape", ["ip-dense", "url-heavy", "skewed-halves", "jsonl-app"])
def test_log_shapes_through_every_device_entry(M, oracle, shape):
    """The log shapes of `bench.py --log-shape` that are NOT nginx against the oracle at 300 K lines of the C2
    database (100 K indicators), TWO consecutive batches through one scanner (the second batch runs with the grids and the
    chunk sizes the first one's list lengths chose: `SparseWriter` chunks of 64 -> 512 slots, sentinel padding), then the
    second batch through every device-resident entry — forked, sliced, submitted, compact records."""
    from tools import synth
    from tests.test_gpu_parity import _device_entries
    cfg = synth.config("c2")
    blob = synth.build_db(cfg)
    db = M.Database(blob)
    sc = M.Scanner(db)
    odb = oracle.Database(blob)
    threads = min(len(os.sched_getaffinity(0)), 16)
    last = None
    for bi in range(2):
        log = synth.make_log(cfg, 5_000_000 + bi * 300000, 300000, shape=shape)
        want, _, st = odb.scan(log, threads=threads, cache=0, want_json=False)
        res = sc.scan(log)
        assert (res.lines, res.candidates) == (st.lines, st.candidates), (shape, bi)
        assert res.hits() == want, (shape, bi)
        res.close()
        last = (log, want, (st.lines, st.candidates))
    assert len(last[1]) > 1000
    _device_entries(sc, last[0], last[1], None, last[2], slices=(3,))
    sc.close(); db.close()


def test_recorded_limits_of_the_hit_record_fail_cleanly(M, oracle):
    """DESIGN §7 deviation (2): the 16-byte hit record holds a candidate's length in 24 bits and its pattern count in 16. A candidate of 16 MiB
    or more (the reference extracts it: nothing bounds an e-mail's local part) makes the call FAIL with a message — never a truncated record —
    and the handle keeps working; just below the limit the item is extracted like the oracle's. (The other limit — a candidate that matches
    more than 65 535 patterns at once — has NO test on purpose: 66 000 backtracking globs against one 500-byte name are 66 000 x up to 100 000
    matcher steps in ONE lane of the spill pass; the round-5 attempt to run exactly that took a GPU box down. DESIGN §7.)"""
    ex = M.Extractor()
    small = b"ab@c.io 1.2.3.4 "
    big = b"x" * ((16 << 20) + 5) + b"@a.com "
    with pytest.raises(RuntimeError) as e:
        ex.extract_from_chunk(small + big)
    assert "16 MiB" in str(e.value), str(e.value)
    assert [(t, v) for t, _, _, v in ex.extract_from_chunk(small)] == [(t, v) for t, _, _, v in oracle.extract(small)]
    under = b"x" * ((16 << 20) - 64) + b"@a.com "
    got, want = ex.extract_from_chunk(under), oracle.extract(under)
    assert [(t, s, e2) for t, s, e2, _ in got] == [(t, s, e2) for t, s, e2, _ in want] and got
    ex.close()
