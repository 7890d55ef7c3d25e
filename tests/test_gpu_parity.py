"""GPU parity tests (-m gpu): the HIP path, called through the C ABI (libmatchy_amd.so), against the CPU oracle on the
same inputs. Integer/byte work: every comparison is exact."""
import json
import os
import random
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu

GOLD = Path(__file__).parent / "golden"
ROOT = Path(__file__).resolve().parent.parent


@pytest.fixture(scope="module")
def M():
    import matchy_amd
    matchy_amd.lib()
    return matchy_amd


@pytest.fixture(scope="module")
def gpu_extractor(M):
    ex = M.Extractor()
    yield ex
    ex.close()


def norm(items):
    return [(t, s, e, v) for (t, s, e, v) in items]


def test_extractor_golden_vectors(M, gpu_extractor, oracle):
    cases = json.loads((GOLD / "extractor_kat.json").read_text())["cases"]
    exs = {}
    for c in cases:
        data = bytes.fromhex(c["input_hex"]) if "input_hex" in c else c["input"].encode()
        key = (c["flags"], c["min_labels"])   # min_labels != 2: matchy_amd_extractor_create(flags, min_domain_labels)
        if key not in exs:
            exs[key] = M.Extractor(flags=c["flags"], min_domain_labels=c["min_labels"])
        got = exs[key].extract_from_chunk(data)
        want = oracle.extract(data, flags=c["flags"], min_labels=c["min_labels"])
        assert norm(got) == norm(want), (c["ref"], data)
        by_type = {}
        for t, s, e, v in got:
            by_type.setdefault(t, []).append(v)
        for t, w in c["expect"].items():
            assert by_type.get(t, []) == w, (c["ref"], t)
    for e in exs.values():
        e.close()


ADVERSARIAL = [
    b"", b"\n", b".", b"..", b"a.b", b"1.2.3.4", b"1.2.3.4\n", b" 1.2.3.4", b"1.2.3.4.", b".1.2.3.4", b"1.2.3.4.5", b"01.2.3.4", b"1.2.3.04",
    b"0.0.0.0", b"255.255.255.255", b"256.1.1.1", b"1.2.3.4:80", b"x1.2.3.4", b"1.2.3.4x", b"[1.2.3.4]", b"1..2.3.4", b"1.2.3.4/24",
    b"999.1.1.1 1.1.1.1", b"a=1.2.3.4;b=5.6.7.8", b"1234.1.1.1", b"1.1.1.1111", b"1.2.3", b"1.2.3.4 5.6.7.8 9.10.11.12",
    b"foo.com", b"foo.com.", b".foo.com", b"foo..com", b"-foo.com", b"foo-.com", b"foo.-com", b"foo.com-", b"a.b.c.d.e.f.com", b"foo.co.uk",
    b"foo.co.za", b"foo.za", b"foo.ck", b"x.foo.ck", b"EXAMPLE.COM", b"example.COM", b"foo_bar.com", b"foo.com_bar", b"foo.com/path", b"http://foo.com:8080/",
    b"user@foo.com", b"a@b", b"@foo.com", b"user@", b"user@@foo.com", b"u.s.e.r@foo.co.uk", b"u..r@foo.com", b"123@foo.com", b"user@foo.com.", b"<user@foo.com>",
    b"a::b", b"::1", b"fe80::1", b"FE80::1", b"feb0::1234:5678", b"fec0::1234:5678", b"2001:db8::1", b"2001:db8::1::2", b"2001:db8:::1", b"1:2:3:4:5:6:7::8",
    b"1:2:3:4:5:6::7:8", b"12345::abcd:1", b"abcd::12345:1", b"x2001:db8::1", b"2001:db8::1x", b"2001:db8::1]", b"::ffff:1.2.3.4", b"2001:db8::", b"::2001:db8:1",
    b"a" * 300 + b".com", b"a." * 200 + b"com", b"1." * 100, b":" * 50, b"::" * 30 + b"1", b"@" * 40, b"." * 100, b"0x" * 30,
    b"5d41402abc4b2a76b9719d911017c592", b"x5d41402abc4b2a76b9719d911017c592", b"5d41402abc4b2a76b9719d911017c592.", b"=5d41402abc4b2a76b9719d911017c592;",
    b"5d41402abc4b2a76b9719d911017c59", b"5d41402abc4b2a76b9719d911017c5922", b"A" * 40, b"a" * 64, b"f" * 96, b"0" * 128, b"0" * 129, b"g" * 32,
    b"0x5aAeb6053F3E94C9b9A09f33669435E7Ef1BeAed", b"0x5aaeb6053f3e94c9b9a09f33669435e7ef1beaed", b"0X5aaeb6053f3e94c9b9a09f33669435e7ef1beaed",
    b"0x5AAEB6053F3E94C9B9A09F33669435E7EF1BEAED", b"x0x5aaeb6053f3e94c9b9a09f33669435e7ef1beaed", b"0x5aaeb6053f3e94c9b9a09f33669435e7ef1beaed0",
    b"1A1zP1eP5QGefi2DMPTfTL5SLmv7DivfNa", b"1A1zP1eP5QGefi2DMPTfTL5SLmv7DivfNb", b"3Cbq7aT1tY8kMxWLbitaG7yT6bPbKChq64", b"bc1qar0srrr7xfkvy5l643lydnw9re59gtzzwf5mdq",
    b"bc1qar0srrr7xfkvy5l643lydnw9re59gtzzwf5mdQ", b"BC1QAR0SRRR7XFKVY5L643LYDNW9RE59GTZZWF5MDQ", b"bc1pw508d6qejxtdg4y5r3zarvary0c5xw7kw508d6qejxtdg4y5r3zarvary0c5xw7kt5nd6y",
    b"bc1p5cyxnuxmeuwuvkwfem96lqzszd02n6xdcjrs20cac6yqjjwudpxqkedrcr", "münchen.de".encode(), "x.münchen.de y".encode(), b"caf\xc3\xa9.fr", b"caf\xc3.fr", b"\xff\xfe.com",
    b"\x80abc.com", b"abc\x80.com", b"a.b\xe2\x82\xac.com", b"a.\xed\xa0\x80.com", b"key=val.ue.com&x=1", b"https://a.b.example.org/p?q=1.2.3.4&r=x@y.io",
]


def test_extractor_adversarial_inputs(gpu_extractor, oracle):
    for data in ADVERSARIAL:
        for wrap in (lambda d: d, lambda d: b"x " + d + b" y", lambda d: d + b"\n" + d, lambda d: b"\n" + d + b"\n"):
            buf = wrap(data)
            assert norm(gpu_extractor.extract_from_chunk(buf)) == norm(oracle.extract(buf)), buf


ALPHABETS = [
    b"0123456789.", b"0123456789abcdef:", b"abc.-", b"ab.@-_+1", b"0x123abcdefABCDEF ", b"a1.:@ /-\n", bytes(range(256)),
    b"comnetorg.uk.co.za", b" \t\n/,;:()[]{}<>\"'@=ab1.", b"13bc1qpzry9x8gf2tvdw0s3jn54khce6mua7l ",
]


def test_extractor_differential_fuzz(gpu_extractor, oracle):
    rng = random.Random(20251212)
    for it in range(400):
        alpha = rng.choice(ALPHABETS)
        n = rng.choice([1, 5, 17, 63, 64, 65, 127, 128, 129, 200, 1000, 1023, 1024, 1025, 5000])
        buf = bytes(rng.choice(alpha) for _ in range(n))
        got, want = norm(gpu_extractor.extract_from_chunk(buf)), norm(oracle.extract(buf))
        assert got == want, (it, buf)


def test_extractor_segment_and_block_edges(gpu_extractor, oracle):
    # tokens / runs that straddle the 64-byte row, 1 KiB block and 16 KiB segment edges of the tokenizer
    payloads = [b"10.20.30.40", b"evil.example.com", b"user@mail.example.org", b"2001:db8::8a2e:370:7334",
                b"5d41402abc4b2a76b9719d911017c592", b"2c26b46b68ffc68ff99b453c1d30413413422d706483bfa0f98a5e886266e7ae",
                b"0x5aAeb6053F3E94C9b9A09f33669435E7Ef1BeAed", b"1A1zP1eP5QGefi2DMPTfTL5SLmv7DivfNa"]
    for edge in (64, 1024, 16384, 32768):
        for pl in payloads:
            for shift in range(0, len(pl) + 2):
                pre = edge - shift
                if pre < 1:
                    continue
                buf = b"a" * (pre - 1) + b" " + pl + b" tail\n"
                assert norm(gpu_extractor.extract_from_chunk(buf)) == norm(oracle.extract(buf)), (edge, pl, shift)
    # unterminated last token ends exactly at the end of the buffer
    for pl in payloads:
        for pad in (0, 1, 15, 16, 63, 64, 1023):
            buf = b"x" * pad + b" " + pl
            assert norm(gpu_extractor.extract_from_chunk(buf)) == norm(oracle.extract(buf)), (pl, pad)


def _device_entries(sc, text, got_hits, got_lines, got_stats, slices=(2, 5), compact_fits=True):
    """The same bytes through the device-resident entries of an existing scanner — forked (independent parts of a scan on several
    streams), sliced, submitted / waited, and with compact IPv4 records — against the results of the host-buffer entry."""
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    dptr = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dptr), ctypes.c_size_t(len(text) + 64)) == 0
    assert hip.hipMemcpy(dptr, bytes(text), ctypes.c_size_t(len(text)), 1) == 0
    try:
        rd = sc.scan_device(dptr.value, len(text), fetch_mode=3)
        assert (rd.lines, rd.candidates) == got_stats
        assert rd.hits() == got_hits
        rd.close()
        # ... and cut into slices (the tail of one slice beside the streaming pass of the next; cuts on 8 KiB multiples, not
        # on newlines): same counters, same records
        for ns in slices:
            if len(text) >= ns * 8192:
                sc.set_slices(ns)
                rs = sc.scan_device(dptr.value, len(text), fetch_mode=3)
                assert sc.last_slices() == ns
                assert (rs.lines, rs.candidates) == got_stats
                assert rs.hits() == got_hits
                rs.close()
        sc.set_slices(0)
        # ... and with the IPv4 results as compact 8-byte records (fetch_mode 1 | 8, device order): same matches, every IPv4 result
        # in the compact array (these databases' data sections are far below the 4 MiB the record addresses); forked, sliced and
        # through submit / wait (one stream)
        key = lambda h: (h["start"], h["end"], h["type"], h["kind"])
        # (a data section beyond 4 MiB keeps the 16-byte records: the flag is ignored, n_ip4_hits 0 — test_compact_records_need_a_small_data_section)
        n_v4 = sum(1 for h in got_hits if h["type"] == "IPv4") if compact_fits else 0
        want_sorted = sorted(got_hits, key=key)
        for how in ("forked", "sliced", "submitted"):
            if how == "sliced":
                if len(text) < 3 * 8192:
                    continue
                sc.set_slices(3)
            if how == "submitted":
                sc.submit_device(dptr.value, len(text), fetch_mode=9)
                rc = sc.wait()
            else:
                rc = sc.scan_device(dptr.value, len(text), fetch_mode=9)
            sc.set_slices(0)
            assert (rc.lines, rc.candidates, rc.n_hits, rc.n_ip4_hits) == (*got_stats, len(got_hits), n_v4), how
            assert sorted(rc.hits(), key=key) == want_sorted, how
            if got_lines is not None:
                assert sorted(rc.ndjson(text, source="t.log")) == sorted(got_lines), how
            rc.close()
    finally:
        sc.set_slices(0)
        hip.hipFree(dptr)


def _scan_both(M, oracle, blob, text, compact_fits=True):
    db = M.Database(blob)
    sc = M.Scanner(db)
    res = sc.scan(text)
    got_hits = res.hits()
    got_lines = res.ndjson(text, source="t.log")
    # the batch renderer (one call, cached data payloads: what `matchy match` prints) gives the same lines, twice in a row
    for _ in range(2):
        assert res.ndjson_text(text, source="t.log") == "".join(l + "\n" for l in got_lines).encode()
    got_stats = (res.lines, res.candidates)
    res.close()
    if text:
        _device_entries(sc, text, got_hits, got_lines, got_stats, compact_fits=compact_fits)
    sc.close(); db.close()
    odb = oracle.Database(blob)
    want_hits, want_lines, st = odb.scan(text, source="t.log")
    return got_hits, got_lines, got_stats, want_hits, want_lines, (st.lines, st.candidates)


def test_scan_small_combined_database(M, oracle):
    b = M.DatabaseBuilder(build_epoch=1)
    for k, v in [("8.8.8.8", {"who": "dns"}), ("evil.com", {"why": "bad"}), ("*.malware.com", {"why": "glob"}), ("10.0.0.0/8", {"net": "rfc1918"}),
                 ("2001:db8::/32", {"net": "doc"}), ("5d41402abc4b2a76b9719d911017c592", {"h": "md5"}), ("glob:paypa1", {"sub": "string"}),
                 ("*.evil.*", {"g": 2}), ("bad-??.example.[a-c]om", {"g": 3}), ("*", {"g": "all"}) if False else ("zzz.invalid", {})]:
        b.add_entry(k, v)
    blob = b.build()
    text = (b"DNS query to evil.com from 8.8.8.8\nGET http://bad.malware.com/x from 10.1.2.3\n"
            b"v6 client 2001:db8:85a3::8a2e:370:7334 fetched www.paypa1-login.net/a?h=5d41402abc4b2a76b9719d911017c592\n"
            b"a.evil.b c.evil.com bad-ab.example.com bad-ab.example.dom nothing.here.zz 11.1.1.1\n")
    gh, gl, gs, wh, wl, ws = _scan_both(M, oracle, blob, text)
    assert gh == wh
    assert gl == wl
    assert gs == ws
    assert len(gh) >= 8


def test_compact_records_need_a_small_data_section(M, oracle):
    """MATCHY_SCAN_FETCH_COMPACT addresses the data section with 22 bits: a database with more than 4 MiB of entry data keeps
    the 16-byte records (the flag is ignored: n_ip4_hits 0, same matches)."""
    import ctypes
    b = M.DatabaseBuilder(build_epoch=1)
    for k in range(40):
        b.add_entry(f"10.{k}.0.0/16", {"blob": ("%02d" % k) * 70000})   # 140 KB of data each: 5.6 MB
    b.add_entry("evil.example.com", {"d": 1})
    blob = b.build()
    text = b"".join(b"client 10.%d.%d.7 asked evil.example.com\n" % (k % 50, k) for k in range(400))
    want, want_lines, st = oracle.Database(blob).scan(text, source="t.log")
    db = M.Database(blob)
    sc = M.Scanner(db)
    hip = ctypes.CDLL("libamdhip64.so")
    dptr = ctypes.c_void_p()
    assert hip.hipMalloc(ctypes.byref(dptr), ctypes.c_size_t(len(text) + 64)) == 0
    assert hip.hipMemcpy(dptr, text, ctypes.c_size_t(len(text)), 1) == 0
    r = sc.scan_device(dptr.value, len(text), fetch_mode=9)
    assert r.n_ip4_hits == 0 and r.n_hits == len(want) > 400
    key = lambda h: (h["start"], h["end"], h["type"])
    assert sorted(r.hits(), key=key) == sorted(want, key=key)
    assert max(h["ip_data_offset"] for h in want) >= 1 << 22
    assert sorted(r.ndjson(text, source="t.log")) == sorted(want_lines)
    r.close()
    hip.hipFree(dptr)
    sc.close(); db.close()


@pytest.mark.parametrize("cfgname,lines", [("c1", 10000), ("c2/20", 20000), ("c3/20", 20000), ("c3b/20", 20000), ("c4/20", 20000),
                                           ("c5/100", 20000)])
def test_scan_synthetic_configs(M, oracle, cfgname, lines):
    from tools import synth
    cfg = synth.config(cfgname)
    blob = synth.build_db(cfg)
    text = synth.make_log(cfg, 0, lines)
    gh, gl, gs, wh, wl, ws = _scan_both(M, oracle, blob, text)
    assert gs == ws
    assert len(gh) == len(wh)
    assert gh == wh
    assert gl == wl
    assert len(gh) > 50


@pytest.mark.parametrize("bits", [28, 32])
def test_scan_wide_tree_records(M, oracle, monkeypatch, bits):
    """BASELINE configs[4] selects 28-bit tree records at full size (18M nodes); here the same indicator mix, scaled, is
    written with 28- and 32-bit records (builder test hook) and scanned: hits and prefix lengths equal the oracle's."""
    from tools import synth
    monkeypatch.setenv("MATCHY_AMD_MIN_RECORD_SIZE", str(bits))
    cfg = synth.config("c5/200")
    blob = synth.build_db(cfg)
    monkeypatch.delenv("MATCHY_AMD_MIN_RECORD_SIZE")
    odb = oracle.Database(blob)
    assert odb.metadata()["record_size"] == bits
    text = synth.make_log(cfg, 0, 10000)
    gh, gl, gs, wh, wl, ws = _scan_both(M, oracle, blob, text)
    assert gs == ws and gh == wh and gl == wl
    assert len(gh) > 50


@pytest.mark.parametrize("cfgname,lines", [("c1", 10000), ("c2", 1000), ("c3", 1000), ("c3b", 1000), ("c4", 1000), ("c5/100", 1000)])
def test_scan_matches_committed_golden(M, cfgname, lines):
    """GPU scan against the committed match sets of the BASELINE configs (tests/golden/config_*.ndjson)."""
    import json
    from pathlib import Path
    from tools import synth
    rows = (Path(__file__).parent / "golden" / f"config_{cfgname.replace('/', '_')}.ndjson").read_text().splitlines()
    head, want = json.loads(rows[0]), rows[1:]
    cfg = synth.config(cfgname)
    db = M.Database(synth.build_db(cfg))
    sc = M.Scanner(db)
    text = synth.make_log(cfg, 0, lines)
    res = sc.scan(text)
    got = res.ndjson(text, source="access.log")
    stats = (res.lines, res.candidates, len(got))
    res.close(); sc.close(); db.close()
    assert stats == (head["lines"], head["candidates"], head["matches"])
    assert got == want


def test_single_query_api(M, oracle):
    from tools import synth
    cfg = synth.config("c1")
    blob = synth.build_db(cfg)
    db = M.Database(blob)
    odb = oracle.Database(blob)
    keys = [k.decode() for k, _ in synth.ioc_entries(cfg)]
    probes = keys[::37] + ["1.2.3.4", "nope.example.com", "::1", "2001:db8::1", "not an ip", ""]
    for k in keys[::37]:
        if "/" in k:
            probes.append(k.split("/")[0])
        if k.startswith("*."):
            probes.append("www" + k[1:])
    for q in probes:
        want = odb.lookup(q)
        got = db.lookup(q)
        if want["kind"] == "ip":
            assert got == {"found": True, "prefix_len": want["prefix_len"], "data": want["data"]}, q
        elif want["kind"] == "pattern" and want["data"] and want["data"][0] is not None:
            assert got == {"found": True, "prefix_len": 0, "data": want["data"][0]}, q
        else:
            assert got is None, q
    db.close()


def test_null_and_error_handling(M):
    L = M.lib()
    assert L.matchy_open(None) is None
    assert L.matchy_open(b"/nonexistent/file.mxy") is None
    assert L.matchy_open_buffer(b"garbage-not-a-db", 16) is None
    assert L.matchy_open_with_options(b"/nonexistent", None) is None
    r = L.matchy_query(None, b"x")
    assert not r.found
    L.matchy_close(None)
    L.matchy_scanner_free(None)


def test_concurrent_queries_and_scans(M, oracle):
    """One database handle shared by several threads (c-querying.md: handles are thread-safe for concurrent queries);
    every thread also owns a scanner. ctypes releases the GIL during the calls, so the C side really runs in parallel."""
    import threading
    from tools import synth
    cfg = synth.config("c1")
    blob = synth.build_db(cfg)
    db = M.Database(blob)
    odb = oracle.Database(blob)
    keys = [k.decode() for k, _ in synth.ioc_entries(cfg)][::9]
    probes = [k.split("/")[0] if "/" in k else ("www" + k[1:] if k.startswith("*.") else k) for k in keys] + ["9.9.9.9", "nope.invalid.example"]
    want_q = {}
    for q in probes:
        w = odb.lookup(q)
        want_q[q] = (w["kind"] != "notfound") and not (w["kind"] == "pattern" and (not w["data"] or w["data"][0] is None))
    logs = [synth.make_log(cfg, 3000 * t, 3000) for t in range(4)]
    want_s = [odb.scan(l, want_json=False)[0] for l in logs]
    errors = []

    def worker(t):
        try:
            sc = M.Scanner(db)
            for rep in range(3):
                for q in probes:
                    got = db.lookup(q)
                    if (got is not None) != want_q[q]:
                        errors.append((t, "query", q))
                r = sc.scan(logs[t])
                if r.hits() != want_s[t]:
                    errors.append((t, "scan", rep))
                r.close()
            sc.close()
        except Exception as e:  # noqa: BLE001
            errors.append((t, "exception", repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(4)]
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    assert not errors, errors[:5]
    db.close()


def test_concurrent_forked_device_scans(M, oracle):
    """Four threads, each with its OWN scanner, twenty device-resident scans apiece at the same time (the reference's workers
    run concurrently: processing/parallel.rs:594-704). `matchy_scanner_scan_device` always forks — side streams per scanner, and a
    k_finish that polls a counter bumped by kernels in other hardware queues — so several forked scans in flight share the
    runtime's few hardware queues: every scan must still return exactly the oracle's records, and quickly (a k_finish that
    starves a chain it is waiting for would sit out its poll time-out: seconds per scan)."""
    import ctypes
    import threading
    import time
    from tools import synth
    cfg = synth.config("c4/10")   # globs: the early glob pass and the undecided domains' chain are side streams too
    blob = synth.build_db(cfg)
    db = M.Database(blob)
    odb = oracle.Database(blob)
    hip = ctypes.CDLL("libamdhip64.so")
    n_threads, n_scans, n_batches = 4, 20, 5
    logs = [[synth.make_log(cfg, 40000 * (t * n_batches + b), 20000 + 1000 * b) for b in range(n_batches)] for t in range(n_threads)]
    want = [[odb.scan(l, want_json=False) for l in per] for per in logs]
    dptrs = [[ctypes.c_void_p() for _ in per] for per in logs]
    for per, dp in zip(logs, dptrs):
        for l, d in zip(per, dp):
            assert hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(len(l) + 64)) == 0
            assert hip.hipMemcpy(d, l, ctypes.c_size_t(len(l)), 1) == 0
    errors = []
    start = threading.Barrier(n_threads)

    def worker(t):
        try:
            sc = M.Scanner(db)
            start.wait(timeout=60)
            for k in range(n_scans):
                b = k % n_batches
                r = sc.scan_device(dptrs[t][b].value, len(logs[t][b]), fetch_mode=3)
                w_hits, _, st = want[t][b]
                if (r.lines, r.candidates) != (st.lines, st.candidates) or r.hits() != w_hits:
                    errors.append((t, k, "records differ"))
                r.close()
            sc.close()
        except Exception as e:  # noqa: BLE001
            errors.append((t, "exception", repr(e)))

    threads = [threading.Thread(target=worker, args=(t,)) for t in range(n_threads)]
    t0 = time.time()
    for th in threads:
        th.start()
    for th in threads:
        th.join(timeout=300)
    took = time.time() - t0
    for dp in dptrs:
        for d in dp:
            hip.hipFree(d)
    db.close()
    assert not errors, errors[:5]
    assert not any(th.is_alive() for th in threads)
    # 80 scans of ~4 MB: a few milliseconds each even when the streams of four forked scans collide in the hardware queues
    assert took < 20.0, took


def test_finish_poll_gives_the_join_back_to_the_host(M, oracle):
    """k_finish polls for the side chains of a forked scan; when it gives up (here: at once, MATCHY_AMD_FINISH_POLL_US=0 in a child
    process) the scan must not fail or lose records: the host joins the side streams and takes the counters again."""
    import subprocess
    import sys
    code = r"""
import ctypes, sys
sys.path.insert(0, %r)
import matchy_amd as M
from tools import synth
from oracle import oracle as orc
orc.build()
cfg = synth.config("c4/10")
blob = synth.build_db(cfg)
log = synth.make_log(cfg, 0, 30000)
db = M.Database(blob); sc = M.Scanner(db)
hip = ctypes.CDLL("libamdhip64.so")
d = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(len(log) + 64)) == 0
assert hip.hipMemcpy(d, log, ctypes.c_size_t(len(log)), 1) == 0
want, _, st = orc.Database(blob).scan(log, want_json=False)
for rep in range(5):
    r = sc.scan_device(d.value, len(log), fetch_mode=3)
    assert (r.lines, r.candidates) == (st.lines, st.candidates), rep
    assert r.hits() == want, rep
    r.close()
print("OK")
""" % str(ROOT)
    env = dict(os.environ, MATCHY_AMD_FINISH_POLL_US="0", MATCHY_AMD_TRACE="1")
    p = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "OK" in p.stdout, p.stderr[-2000:]
    assert "joining the side streams on the host" in p.stderr, p.stderr[-2000:]


def test_long_tokens_beside_the_rare_anchors(M, oracle):
    """When the previous batch listed hundreds of thousands of long tokens AND of IPv6 / e-mail anchors, the engine runs the token
    pass beside the rare-anchor pass (second stream, in front of k_rare, its hashes in k_rare's candidate list) instead of in front
    of it. Forced on in a child process (MATCHY_AMD_TOK_ASIDE=1) over a JSON-lines batch, which has both kinds, and — so that the
    adaptive switch itself runs — left to the engine over the same batch scanned twice (the second scan sees the first one's lists)."""
    import subprocess
    import sys
    code = r"""
import ctypes, sys
sys.path.insert(0, %r)
import matchy_amd as M
from tools import synth
from oracle import oracle as orc
orc.build()
cfg = synth.config("c2/10")
blob = synth.build_db(cfg)
log = synth.make_log(cfg, 0, int(sys.argv[1]), shape="jsonl-app")
db = M.Database(blob); sc = M.Scanner(db)
hip = ctypes.CDLL("libamdhip64.so")
d = ctypes.c_void_p()
assert hip.hipMalloc(ctypes.byref(d), ctypes.c_size_t(len(log) + 64)) == 0
assert hip.hipMemcpy(d, log, ctypes.c_size_t(len(log)), 1) == 0
want, _, st = orc.Database(blob).scan(log, want_json=False)
assert len(want) > 100
for rep in range(3):
    r = sc.scan_device(d.value, len(log), fetch_mode=3)
    assert (r.lines, r.candidates) == (st.lines, st.candidates), rep
    assert r.hits() == want, rep
    r.close()
print("OK")
""" % str(ROOT)
    # forced: a small batch; adaptive: 800 K lines list 990 K long tokens and 310 K rare anchors, above the switch (262 144 of each)
    for setting, lines in (("1", 120000), (None, 800000)):
        env = dict(os.environ)
        env.pop("MATCHY_AMD_TOK_ASIDE", None)
        if setting is not None:
            env["MATCHY_AMD_TOK_ASIDE"] = setting
        p = subprocess.run([sys.executable, "-c", code, str(lines)], env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "OK" in p.stdout, (setting, p.stderr[-2000:])


def test_multi_device_scanner_matches_the_single_scanner(M, oracle, tmp_path):
    """matchy_multi_scanner_*: the reader -> per-device workers -> ordered gather behind the C ABI (processing/parallel.rs:494-505).
    The same GPU listed three times: a buffer (merged result = the oracle's records with absolute offsets, for several piece sizes),
    a file (batches in file order, offsets rebased by the batch's position) and a pipe."""
    from tools import synth
    cfg = synth.config("c4/10")
    blob = synth.build_db(cfg)
    db = M.Database(blob)
    odb = oracle.Database(blob)
    log = synth.make_log(cfg, 0, 60000)
    want, _, st = odb.scan(log, want_json=False)
    assert len(want) > 500
    ms = M.MultiScanner(db, devices=(0, 0, 0))
    assert ms.workers == 3
    for bb in (0, 1 << 20, 300000, len(log) * 2):
        r = ms.scan(log, batch_bytes=bb)
        assert (r.lines, r.candidates) == (st.lines, st.candidates), bb
        assert r.hits() == want, bb
        r.close()
    r = ms.scan(b"")
    assert r.hits() == [] and r.lines == 0
    r.close()
    # a file: batches come back in file order whatever worker scanned them
    path = tmp_path / "access.log"
    path.write_bytes(log)
    got, offsets = [], []

    def on_batch(off, n, hits, lines, cands):
        offsets.append((off, n))
        for h in hits:
            h = dict(h); h["start"] += off; h["end"] += off
            got.append(h)

    tot = ms.scan_file(str(path), batch_bytes=1 << 20, on_batch=on_batch)
    assert offsets == sorted(offsets) and len(offsets) >= 8 and sum(n for _, n in offsets) == len(log)
    assert all(log[o + n - 1:o + n] == b"\n" for o, n in offsets)
    assert got == want
    assert (tot["lines"], tot["candidates"], tot["matches"], tot["bytes"]) == (st.lines, st.candidates, len(want), len(log))
    # a pipe (read, not mapped)
    import threading
    rd, wr = os.pipe()

    def feed():
        with os.fdopen(wr, "wb") as f:
            f.write(log)

    th = threading.Thread(target=feed)
    th.start()
    got.clear(); offsets.clear()
    tot = ms.scan_file(f"/proc/self/fd/{rd}", batch_bytes=1 << 20, on_batch=on_batch)
    th.join()
    os.close(rd)
    assert got == want and tot["lines"] == st.lines
    # a missing file is an error, not an empty scan
    with pytest.raises(RuntimeError):
        ms.scan_file(str(tmp_path / "nope.log"))
    ms.close(); db.close()


def _long_domain_runs(seed, count=700):
    """Runs of 40..3000 domain characters: many labels, dashes beside dots at every alignment, empty labels, high bytes."""
    rng = random.Random(seed)
    alpha = b"abcdefghijklmnopqrstuvwxyz0123456789"
    tails = [b"com", b"co.uk", b"ck", b"www.ck", b"zip", b"nosuchtld", b"museum", b"xn--p1ai", "рф".encode()]
    buf = bytearray()
    for i in range(count):
        parts = []
        total = 0
        want_len = rng.choice([40, 64, 100, 257, 700, 3000])
        while total < want_len:
            n = rng.choice([1, 2, 3, 7, 8, 9, 15, 16, 17, 31, 63, 64, 120])
            lab = bytearray(rng.choice(alpha) for _ in range(n))
            r = rng.random()
            if r < 0.015:
                lab = bytearray(b"-") + lab
            elif r < 0.03:
                lab += b"-"
            elif r < 0.06 and n > 2:
                lab[rng.randrange(1, n - 1)] = ord("-")
            elif r < 0.07:
                lab = bytearray()
            elif r < 0.09:
                lab += "ü".encode()
            elif r < 0.095:
                lab += b"\xc3"
            parts.append(bytes(lab))
            total += len(lab) + 1
        buf += rng.choice([b" ", b"\n", b"/", b"=", b"x ", b"_", b"@"]) + b".".join(parts + [rng.choice(tails)]) + rng.choice([b" ", b"\n", b"/", b":", b"_"])
    return bytes(buf)


def _long_emails(seed, count=500):
    rng = random.Random(seed)
    alpha = b"abcdefghijklmnopqrstuvwxyz0123456789ABC"
    buf = bytearray()
    for i in range(count):
        n = rng.choice([1, 7, 8, 9, 24, 31, 32, 33, 40, 100, 1000])
        r = rng.random()
        loc = bytearray(rng.choice(b"0123456789") if r < 0.1 else rng.choice(alpha) for _ in range(n))
        for _ in range(rng.choice([0, 0, 1, 3, 10])):
            loc[rng.randrange(n)] = rng.choice(b"._+-.")
        if rng.random() < 0.1 and n > 2:
            k = rng.randrange(n - 1)
            loc[k:k + 2] = b".."
        if rng.random() < 0.05:
            loc[rng.randrange(n)] = rng.choice(b"!\xc3~*")
        dom = b".".join(bytes(rng.choice(alpha) for _ in range(rng.choice([1, 5, 8, 30, 200]))) for _ in range(rng.choice([1, 2, 3, 12])))
        dom += rng.choice([b".com", b".co.uk", b".nosuch", b"", b".ck", b".x.ck", b".museum", b"\xc3\xa9.com"])
        buf += rng.choice([b" ", b"\n", b"<", b"x", b"=", b"\xff"]) + bytes(loc) + b"@" + dom + rng.choice([b" ", b"\n", b">", b"_", b"@"])
    return bytes(buf)


@pytest.mark.parametrize("seed", [1, 2])
def test_long_domain_runs(M, oracle, seed):
    """The word-at-a-time continuation of the backward domain walk (domain_walk_back, after the public-suffix question is
    settled) against the oracle's byte walk."""
    buf = _long_domain_runs(seed)
    ex = M.Extractor()
    got, want = ex.extract_from_chunk(buf), oracle.extract(buf)
    ex.close()
    assert sum(1 for t, s, e, v in want if t == "Domain" and e - s > 64) > 100
    assert got == want
    buf = _long_emails(seed)
    ex = M.Extractor()
    got, want = ex.extract_from_chunk(buf), oracle.extract(buf)
    ex.close()
    assert sum(1 for t, s, e, v in want if t == "Email" and e - s > 64) > 50
    assert got == want


@pytest.mark.parametrize("seed", [42, 7, 20251212])
def test_domain_rules_structured_fuzz(M, oracle, seed):
    """Structure-aware differential fuzz of the domain rules (labels with dashes, empty labels, high bytes, long names,
    real and fake TLDs, every kind of delimiter): exercises the mask-arithmetic fast path of k_validate_dom, its hand-over
    to the general walk, and the literal bitmap (scan against a database that contains some of the generated names)."""
    rng = random.Random(seed)
    tlds = [b"com", b"net", b"org", b"io", b"co.uk", b"uk", b"ck", b"www.ck", b"za", b"co.za", b"html", b"zip", b"museum", b"photography",
            b"xn--p1ai", b"a", b"comx", b"c-m", b"", "рф".encode(), b"COM", b"travel", b"b\xc3\xbccher"]
    alpha = b"abcdefghijklmnopqrstuvwxyz0123456789"
    delims = [b" ", b"\n", b"/", b",", b"(", b")", b"\"", b"=", b":", b";", b"<", b">", b"[", b"x", b"_", b"-", b".", b"", b"\t", b"'", b"@", b"\xff"]

    def label():
        r = rng.random()
        n = rng.choice([1, 1, 2, 3, 5, 8, 13, 24, 40])
        s = bytes(rng.choice(alpha) for _ in range(n))
        if r < 0.08:
            s = b"-" + s
        elif r < 0.16:
            s = s + b"-"
        elif r < 0.22:
            s = s[: n // 2] + b"-" + s[n // 2:]
        elif r < 0.26:
            s = b""
        elif r < 0.30:
            s = s + "é".encode()
        elif r < 0.32:
            s = s + b"\xc3"          # truncated UTF-8
        elif r < 0.36:
            s = s.upper()
        return s

    names = []
    for _ in range(6000):
        k = rng.choice([0, 1, 1, 2, 2, 3, 5])
        parts = [label() for _ in range(k)] + [rng.choice(tlds)]
        names.append(b".".join(parts))
    buf = bytearray()
    for nm in names:
        buf += rng.choice(delims) + nm + rng.choice(delims)
        if rng.random() < 0.1:
            buf += b"q" * rng.choice([1, 7, 23, 24, 25, 100])
    buf = bytes(buf)
    ex = M.Extractor()
    got, want = ex.extract_from_chunk(buf), oracle.extract(buf)
    ex.close()
    assert len(got) == len(want)
    assert got == want
    # lookup path: database with a sample of the valid names (+ one IP so that the trie exists)
    valid = sorted({v for t, s, e, v in want if t == "Domain"})
    assert len(valid) > 300
    for ci in (False, True):   # case-insensitive: the bitmap test of the general-walk names folds ASCII and leaves non-ASCII / long names listed
        b = M.DatabaseBuilder(build_epoch=3, case_insensitive=ci)
        for v in valid[::5]:
            b.add_entry("literal:" + v, {"n": len(v)})
        b.add_entry("192.0.2.1", {"ip": True})
        blob = b.build()
        gh, gl, gs, wh, wl, ws = _scan_both(M, oracle, blob, buf)
        assert gs == ws and gh == wh and gl == wl
        assert len(gh) >= len(valid[::5])


@pytest.mark.parametrize("seed", [1, 2])
def test_ipv4_structured_fuzz(M, oracle, seed):
    """Dotted-quad look-alikes with every kind of octet (leading zeros, > 255, 1-4 digits, missing / extra octets) and
    delimiter, packed densely so that windows straddle the kernel's blocks; then scanned against a CIDR-heavy database
    (exercises the first-level table, the /24 bitmap and the prefix-length quirk)."""
    rng = random.Random(seed)
    delims = [b" ", b"\n", b"/", b",", b":", b"[", b"]", b"=", b"x", b".", b"", b"-", b"\"", b"a", b"1"]

    def octet():
        r = rng.random()
        if r < 0.55:
            return str(rng.randrange(256)).encode()
        if r < 0.65:
            return str(rng.randrange(256, 1000)).encode()
        if r < 0.75:
            return b"0" + str(rng.randrange(100)).encode()
        if r < 0.80:
            return b""
        if r < 0.85:
            return str(rng.randrange(1000, 100000)).encode()
        return rng.choice([b"0", b"00", b"255", b"256", b"1", b"10", b"192", b"168"])

    buf = bytearray()
    for _ in range(8000):
        k = rng.choice([2, 3, 4, 4, 4, 4, 5, 6])
        buf += rng.choice(delims) + b".".join(octet() for _ in range(k)) + rng.choice(delims)
    buf = bytes(buf)
    ex = M.Extractor()
    got, want = ex.extract_from_chunk(buf), oracle.extract(buf)
    ex.close()
    assert got == want
    ips = sorted({v for t, s, e, v in want if t == "IPv4"})
    assert len(ips) > 200
    b = M.DatabaseBuilder(build_epoch=4)
    for i, ip in enumerate(ips[::7]):
        a = ip.split(".")
        pfx = rng.choice([32, 32, 24, 16, 27, 12, 30])
        b.add_entry(ip if pfx == 32 else f"{ip}/{pfx}", {"i": i, "p": pfx})
    b.add_entry("0.0.0.0/5", {"wide": True})
    blob = b.build()
    gh, gl, gs, wh, wl, ws = _scan_both(M, oracle, blob, buf)
    assert gs == ws and gh == wh and gl == wl
    assert len(gh) > 40


def _glob_fuzz_case(seed):
    """Names (some longer than the 64-byte text window of the glob pass, some with multi-byte characters) and globs cut out
    of them: stars, question marks, classes, literal pieces shorter and longer than 8 bytes."""
    rng = random.Random(seed)
    alpha = "abcdefghijklmnopqrstuvwxyz0123456789"
    tlds = ["com", "net", "org", "io", "co.uk", "museum"]

    def name():
        k = rng.choice([1, 2, 2, 3, 4, 8])
        labs = []
        for _ in range(k):
            n = rng.choice([1, 2, 3, 5, 8, 9, 13, 21, 40])
            lab = "".join(rng.choice(alpha) for _ in range(n))
            if rng.random() < 0.1 and n > 2:
                lab = lab[:1] + "-" + lab[2:]
            if rng.random() < 0.06:
                lab = lab + "ü"
            labs.append(lab)
        return ".".join(labs + [rng.choice(tlds)])

    names = [name() for _ in range(400)]
    pats = {}
    for _ in range(600):
        nm = rng.choice(names)
        chars = list(nm)
        out = []
        i = 0
        wild = False
        while i < len(chars):
            r = rng.random()
            if r < 0.06:
                out.append("*"); i += rng.choice([0, 1, 3, 10, 30]); wild = True
            elif r < 0.10:
                out.append("?"); i += 1; wild = True
            elif r < 0.13 and chars[i] not in "-.ü":
                c = chars[i]
                out.append(rng.choice([f"[{c}]", f"[!{c}]", "[a-z]", "[0-9]", "[!a-m]", f"[{c}x]"])); i += 1; wild = True
            else:
                out.append(chars[i]); i += 1
        if rng.random() < 0.3:
            out.insert(0, "*"); wild = True
        pt = "".join(out)
        if wild and pt not in pats and pt.strip("*?") != "":
            pats[pt] = len(pats)
    log = bytearray()
    for nm in names:
        log += nm.encode() + rng.choice([b" ", b"\n", b"/", b"\" "])
        if rng.random() < 0.5:   # a near miss
            m = list(nm)
            m[rng.randrange(len(m))] = rng.choice(alpha)
            log += "".join(m).encode() + b"\n"
    return pats, bytes(log)


@pytest.mark.parametrize("seed", [11, 12, 13])
def test_glob_differential_fuzz(M, oracle, seed):
    """Paraglob::find_all on the GPU (AC prefilter in k_validate_dom, DFA walk, glob matcher with its LDS text window and
    cached segment header) against the oracle's restatement of match_segments_impl, over a scan."""
    pats, log = _glob_fuzz_case(seed)
    b = M.DatabaseBuilder(build_epoch=6)
    for pt, i in pats.items():
        b.add_entry(pt, {"g": i})
    b.add_entry("literal:" + log.split()[0].decode(), {"lit": True})
    blob = b.build()
    gh, gl, gs, wh, wl, ws = _scan_both(M, oracle, blob, log)
    assert gs == ws
    assert gh == wh
    assert gl == wl
    assert len(gh) > 100


def _mutated_log(seed, lines, count=400):
    """Lines of a synthetic access log with random byte-level damage: delimiters swapped in, bytes dropped, dots / dashes /
    '@' / '::' / '0x' spliced in, pieces duplicated, raw bytes — candidates that almost parse."""
    rng = random.Random(seed)
    out = bytearray()
    for ln in rng.sample(lines, min(count, len(lines))):
        b = bytearray(ln)
        for _ in range(rng.choice([0, 1, 1, 2, 5])):
            if not b:
                break
            k = rng.randrange(len(b))
            r = rng.random()
            if r < 0.3:
                b[k] = rng.choice(b" ./:@-_=\"'[](){}<>,;\t\xc3\xa9\xff0aZ9")
            elif r < 0.5:
                del b[k]
            elif r < 0.7:
                b[k:k] = rng.choice([b".", b"..", b"-.", b".-", b"@", b"::", b"0x", b" ", b"1.1.1.1", b".com", b"\xe2\x80\x9c", b"a" * 30])
            elif r < 0.8:
                b[k:k] = b[max(0, k - 20):k]
            else:
                b[k] = rng.randrange(256)
        out += b + rng.choice([b"\n", b"\n", b"\r\n", b" ", b""])
    return bytes(out)


@pytest.mark.parametrize("seed", [5, 6])
def test_mutated_log_fuzz(M, oracle, seed):
    """Damaged access-log lines through the extractor and through a scan against the config-4 mix (globs, literals, IPs)."""
    from tools import synth
    cfg = synth.config("c4/50")
    blob = synth.build_db(cfg)
    buf = _mutated_log(seed, synth.make_log(cfg, 0, 3000).split(b"\n"), 1500)
    ex = M.Extractor()
    got, want = norm(ex.extract_from_chunk(buf)), norm(oracle.extract(buf))
    ex.close()
    assert got == want
    gh, gl, gs, wh, wl, ws = _scan_both(M, oracle, blob, buf)
    assert gs == ws and gh == wh and gl == wl


REFERENCE_BEHAVIOUR = [
    # (entries, queries) — the behavioural vectors of the reference's own tests (test_ip_longest_prefix_match.rs,
    # test_literal_hash.rs, paraglob_offset.rs:1890-1944, matchy-paraglob/tests/integration_tests.rs), as in
    # tests/test_builder_oracle.py, here through matchy_query on the GPU lookup kernels
    ([("192.0.2.1", {"type": "specific"}), ("192.0.2.0/24", {"type": "general"})], ["192.0.2.1", "192.0.2.2", "192.0.3.1"]),
    ([("192.0.0.0/8", {"level": "8"}), ("192.0.2.1", {"level": "32"}), ("192.0.2.0/24", {"level": "24"})],
     ["192.0.2.1", "192.0.2.2", "192.1.1.1", "193.0.0.1"]),
    ([("2001:db8::/64", {"level": "64"}), ("2001:db8::1", {"level": "128"}), ("2001:db8::/96", {"level": "96"})],
     ["2001:db8::1", "2001:db8::2", "2001:db8::1:0:0", "2001:db9::1"]),
    ([("2001:db8::1", {"v": 6}), ("10.1.2.3", {"v": 4}), ("10.9.0.0/16", {"v": 16})], ["10.1.2.3", "10.9.77.1", "10.1.2.4", "2001:db8::1"]),
    ([(f"pattern_{i}", {"id": i}) for i in range(100)], ["pattern_0", "pattern_57", "pattern_99", "pattern_100", "pattern_"]),
    ([("*.txt", {"p": 0}), ("test_*", {"p": 1})], ["test_file.txt", "test_file.bin", "other.bin", "x.txt"]),
    ([("*", {"p": 0}), ("??", {"p": 1})], ["ab", "abc", ""]),
    ([("*test*", {"p": 0}), ("test*", {"p": 1}), ("*test", {"p": 2})], ["test", "testing", "mytest", "mytesting", "tes"]),
    ([("glob:hello", {"p": 0}), ("glob:world", {"p": 1})], ["hello world", "say hello", "wor ld"]),
    ([("file[0-9].txt", {"p": 0}), ("file[!0-9].txt", {"p": 1})], ["file7.txt", "fileX.txt", "file.txt", "file77.txt"]),
    ([("*.a?", {"p": 0}), ("*.evil.com", {"p": 1})], ["x.ab", "www.evil.com", "evil.com"]),
    ([("café*.fr", {"u": 1}), ("literal:naïve.example", {"u": 2})], ["café-de-flore.fr", "cafe.fr", "naïve.example", "naive.example"]),
]


def test_reference_behaviour_vectors_through_matchy_query(M, oracle):
    from tests.test_builder_oracle import IP_EXACT_MATCH_KAT
    # test_ip_exact_match.rs: found / not found as the reference asserts it (the loop below also compares with the oracle)
    for ref, entries, found, missing in IP_EXACT_MATCH_KAT:
        b = M.DatabaseBuilder(build_epoch=5)
        for k, v in entries:
            b.add_entry(k, v)
        db = M.Database(b.build())
        for q in found:
            assert db.lookup(q) is not None and db.lookup(q)["found"], (ref, q)
        for q in missing:
            assert db.lookup(q) is None, (ref, q)
        db.close()
    for entries, queries in REFERENCE_BEHAVIOUR + [(e, f + m) for _, e, f, m in IP_EXACT_MATCH_KAT]:
        b = M.DatabaseBuilder(build_epoch=5)
        for k, v in entries:
            b.add_entry(k, v)
        blob = b.build()
        db = M.Database(blob)
        odb = oracle.Database(blob)
        for q in queries:
            want = odb.lookup(q)
            got = db.lookup(q)
            if want["kind"] == "ip":
                assert got == {"found": True, "prefix_len": want["prefix_len"], "data": want["data"]}, (entries[0][0], q)
            elif want["kind"] == "pattern" and want["data"] and want["data"][0] is not None:
                assert got == {"found": True, "prefix_len": 0, "data": want["data"][0]}, (entries[0][0], q)
            else:
                assert got is None, (entries[0][0], q, got)
        db.close()


def test_structured_data_walkers_and_stats(M):
    """matchy_result_get_entry / matchy_aget_value / matchy_get_entry_data_list / matchy_get_stats
    (c_api/matchy.rs:989-1006, 1734-1960; the reference's own coverage: tests/test_c_api_extensions.c)."""
    b = M.DatabaseBuilder(build_epoch=6)
    data = {"country": "US", "asn": 13335, "score": -7, "big": 2 ** 40, "ratio": 0.5, "flag": True,
            "tags": ["a", "bb", {"deep": "x"}], "geo": {"lat": 37.75, "name": "somewhere"}}
    b.add_entry("8.8.8.0/24", data)
    b.add_entry("evil.example", {"kind": "literal"})
    b.add_entry("*.bad.example", {"kind": "glob"})
    db = M.Database(b.build())
    T_STR, T_DBL, T_U16, T_U32, T_MAP, T_I32, T_U64, T_ARR, T_BOOL = 2, 3, 5, 6, 7, 8, 9, 11, 14
    assert db.get_value("8.8.8.8", "country") == (0, T_STR, "US")
    assert db.get_value("8.8.8.8", "asn") == (0, T_U16, 13335)           # serde typing: smallest unsigned that fits
    assert db.get_value("8.8.8.8", "score") == (0, T_I32, -7)
    assert db.get_value("8.8.8.8", "big") == (0, T_U64, 2 ** 40)
    assert db.get_value("8.8.8.8", "ratio") == (0, T_DBL, 0.5)
    assert db.get_value("8.8.8.8", "flag") == (0, T_BOOL, True)
    assert db.get_value("8.8.8.8", "geo") == (0, T_MAP, 2)                # maps / arrays: element count
    assert db.get_value("8.8.8.8", "geo", "name") == (0, T_STR, "somewhere")
    assert db.get_value("8.8.8.8", "tags") == (0, T_ARR, 3)
    assert db.get_value("8.8.8.8", "tags", 1) == (0, T_STR, "bb")
    assert db.get_value("8.8.8.8", "tags", 2, "deep") == (0, T_STR, "x")
    assert db.get_value("8.8.8.8") == (0, T_MAP, 8)                       # empty path: the root
    assert db.get_value("8.8.8.8", "nope")[0] == -7                       # LOOKUP_PATH_INVALID
    assert db.get_value("8.8.8.8", "tags", 3)[0] == -7
    assert db.get_value("8.8.8.8", "tags", "x")[0] == -7
    assert db.get_value("8.8.8.8", "country", "x")[0] == -7               # path through a scalar
    assert db.get_value("9.9.9.9", "country")[0] == -8                    # NO_DATA: not found
    assert db.get_value("evil.example", "kind") == (0, T_STR, "literal")
    assert db.get_value("x.bad.example", "kind") == (0, T_STR, "glob")
    lst = db.entry_data_list("8.8.8.8")
    # node, then children; map values in key order: asn, big, country, flag, geo{lat,name}, ratio, score, tags[...]
    assert lst == [(T_MAP, 8), (T_U16, 13335), (T_U64, 2 ** 40), (T_STR, "US"), (T_BOOL, True), (T_MAP, 2), (T_DBL, 37.75),
                   (T_STR, "somewhere"), (T_DBL, 0.5), (T_I32, -7), (T_ARR, 3), (T_STR, "a"), (T_STR, "bb"), (T_MAP, 1), (T_STR, "x")]
    assert db.entry_data_list("9.9.9.9") is None
    # statistics: Database::lookup accounting (database.rs:725-804). Uncached: a miss counts as a string query (:786-790);
    # from the query cache: a cached miss is typed by parsing the query (:743-752), so "9.9.9.9" is an IP query then
    M.lib().matchy_clear_cache(db.handle)
    st0 = db.stats()
    answers = [db.lookup(q) for q in ("8.8.8.8", "evil.example", "9.9.9.9", "nope.example")]
    st1 = db.stats()
    d = {k: st1[k] - st0[k] for k in st1}
    assert d == {"total_queries": 4, "queries_with_match": 2, "queries_without_match": 2, "cache_hits": 0, "cache_misses": 4,
                 "ip_queries": 1, "string_queries": 3}
    again = [db.lookup(q) for q in ("8.8.8.8", "evil.example", "9.9.9.9", "nope.example")]
    st2 = db.stats()
    d = {k: st2[k] - st1[k] for k in st2}
    assert again == answers
    assert d == {"total_queries": 4, "queries_with_match": 2, "queries_without_match": 2, "cache_hits": 4, "cache_misses": 0,
                 "ip_queries": 2, "string_queries": 2}
    assert M.lib().matchy_has_pattern_data(db.handle) is True
    M.lib().matchy_clear_cache(db.handle)
    db.lookup("8.8.8.8")
    assert db.stats()["cache_misses"] - st2["cache_misses"] == 1
    db.close()


def _ci_case(seed):
    """Indicators and log text in mixed case, with non-ASCII letters (cased and caseless), capital sigmas in final and medial
    position, a dotted capital I, and globs with classes — for a case-insensitive database."""
    rng = random.Random(seed)
    words = ["alpha", "Bravo", "CHARLIE", "délta", "ÉCHO", "foxtrot", "ΓΟΛΦ", "οδυσσευς", "ΟΔΥΣΣΕΥΣ", "İstanbul", "straße", "日本語", "hotel", "India"]
    tlds = ["com", "net", "org", "io", "co.uk"]

    def recase(s):
        r = rng.random()
        if r < 0.25:
            return s.upper()
        if r < 0.5:
            return s.lower()
        if r < 0.75:
            return "".join(ch.upper() if rng.random() < 0.5 else ch.lower() for ch in s)
        return s

    names = []
    for i in range(250):
        k = rng.choice([1, 2, 2, 3])
        names.append(".".join(rng.choice(words) + (str(rng.randrange(50)) if rng.random() < 0.5 else "") for _ in range(k)) + "." + rng.choice(tlds))
    entries = []
    for i, nm in enumerate(names[::3]):
        r = rng.random()
        key = recase(nm)
        if r < 0.5:
            entries.append((key, {"lit": i}))
        elif r < 0.7:
            entries.append(("*." + key.split(".", 1)[1], {"suffix": i}))
        elif r < 0.8:
            entries.append(("glob:" + key.split(".")[0], {"sub": i}))
        elif r < 0.9 and key[0].isascii() and key[0].isalpha():
            entries.append(("[" + key[0] + "x]" + key[1:], {"cls": i}))
        else:
            entries.append((key[:3] + "*" + key[-6:], {"mid": i}))
    seen, uniq = set(), []
    for k, v in entries:
        if k not in seen:
            seen.add(k)
            uniq.append((k, v))
    entries = uniq + [("10.1.0.0/16", {"ip": 1}), ("5D41402ABC4B2A76B9719D911017C592", {"md5": 1}), ("Admin@Example.COM", {"mail": 1})]
    log = bytearray()
    for nm in names:
        log += rng.choice([b"GET http://", b"host=", b"\"", b" "]) + recase(nm).encode() + rng.choice([b"/x ", b"\" ", b" ", b"\n"])
        if rng.random() < 0.2:
            log += b"10.1." + str(rng.randrange(256)).encode() + b".7 5d41402abc4b2a76b9719d911017c592 admin@EXAMPLE.com ADMIN@example.COM\n"
    return entries, bytes(log), names


@pytest.mark.parametrize("seed", [21, 22])
def test_case_insensitive_database(M, oracle, seed):
    """matchy_builder_set_case_insensitive / `matchy build -i`: literal keys and queries go through Rust's to_lowercase (Unicode
    table + Final_Sigma on the device for non-ASCII text, inline ASCII folding otherwise), AC literals are lower-cased and the
    text ASCII-folded, glob literals and classes compare with ASCII folding — scan and single queries against the oracle."""
    entries, log, names = _ci_case(seed)
    b = M.DatabaseBuilder(build_epoch=7, case_insensitive=True)
    for k, v in entries:
        b.add_entry(k, v)
    blob = b.build()
    b.close()
    gh, gl, gs, wh, wl, ws = _scan_both(M, oracle, blob, log)
    assert gs == ws
    assert gh == wh
    assert gl == wl
    assert len(gh) > 60
    assert sum(1 for h in wh if not log[h["start"]:h["end"]].isascii()) > 5
    # the same entries case-sensitively hit less (the mode really matters for this input)
    b2 = M.DatabaseBuilder(build_epoch=7)
    for k, v in entries:
        b2.add_entry(k, v)
    cs_hits, _, _ = oracle.Database(b2.build()).scan(log, want_json=False)
    b2.close()
    assert len(cs_hits) < len(wh)
    # single queries
    db = M.Database(blob)
    odb = oracle.Database(blob)
    rng = random.Random(seed)
    for nm in names[:120] + ["ΟΔΥΣΣΕΥΣ.COM", "οδυσσευς.com", "İSTANBUL.NET", "x" * 300 + "É.com", "STRASSE.ORG", "admin@example.com"]:
        q = "".join(ch.upper() if rng.random() < 0.5 else ch.lower() for ch in nm)
        want, got = odb.lookup(q), db.lookup(q)
        if want["kind"] == "pattern" and want["data"] and want["data"][0] is not None:
            assert got == {"found": True, "prefix_len": 0, "data": want["data"][0]}, q
        elif want["kind"] == "pattern":
            pass   # a hit whose first pattern carries no data: matchy_query reports found=false (c_api/matchy.rs:1143-1154)
        else:
            assert got is None, (q, got)
    db.close()


def test_glob_results_and_star_nesting_beyond_lane_storage(M, oracle):
    """Paraglob::find_all has no cap on results or on '*' nesting (paraglob_offset.rs:1028-1182, 1402-1639). The glob pass of
    k_lookup keeps 32 ids and 24 star frames per lane; a candidate beyond that goes to the spill pass (one bit per pattern id,
    a stack as deep as the longest pattern) and gets the same answer as from the CPU path — scan and single query."""
    b = M.DatabaseBuilder(build_epoch=7)
    name = "alpha.bravo.charlie.delta.echo.foxtrot.golf.hotel.example.com"
    n_over = 0
    # 100 distinct globs that all match `name`
    for i in range(100):
        head = name[: 1 + (i % 40)]
        tail = name[len(name) - 4 - (i // 40) * 3:]
        key = head + "*" + tail if i % 2 else "*" + name[i % 30 + 1:]
        b.add_entry(key + ("" if i < 50 else "*"), {"g": i})
        n_over += 1
    # a pattern with 40 '*' (deeper than the 24 frames a lane of the glob pass holds)
    deep = "*".join(name[k] for k in range(0, 41)) + "*com"
    b.add_entry(deep, {"deep": 1})
    # a SHORT name that more than 32 globs match: k_validate_dom decides it from its context record and flags it for the glob pass itself
    # (the long name above is an "undecided" anchor: the general walk, a candidate list and a glob pass of its own) — both roads spill
    short = "a1.b2c3.example.com"
    for i in range(48):
        b.add_entry(short[: 1 + i % 9] + "*" + short[9 + i // 9:], {"s": i})
    b.add_entry("*.other-suffix.org", {"o": 1})
    b.add_entry("plain.example.net", {"lit": 1})
    blob = b.build()
    b.close()
    log = (b"GET http://" + name.encode() + b"/x 1.2.3.4\n" + b"host=www.other-suffix.org plain.example.net\n") * 3 + b"ref=" + name.encode() + b"\n" + b"short " + short.encode() + b" again " + short.encode() + b"\n"
    gh, gl, gs, wh, wl, ws = _scan_both(M, oracle, blob, log)
    assert gs == ws
    assert gh == wh
    assert gl == wl
    big = [h for h in wh if log[h["start"]:h["end"]] == name.encode()]
    assert len(big) == 4 and all(len(h["ids"]) > 60 for h in big)
    small = [h for h in wh if log[h["start"]:h["end"]] == short.encode()]
    assert len(small) == 2 and all(len(h["ids"]) > 32 for h in small)
    db = M.Database(blob)
    odb = oracle.Database(blob)
    for q in (name, short, "www.other-suffix.org", "plain.example.net", "nothing.example.org"):
        want, got = odb.lookup(q), db.lookup(q)
        if want["kind"] == "pattern" and want["data"] and want["data"][0] is not None:
            assert got == {"found": True, "prefix_len": 0, "data": want["data"][0]}, q
        else:
            assert got is None, (q, got)
    db.close()


def test_case_insensitive_long_non_ascii_key(M, oracle):
    """A case-insensitive database with a non-ASCII literal key far longer than 256 bytes (the device lower-cases the query as a
    stream, no buffer limits it): lit:467-525 with to_lowercase, scan and single query against the oracle."""
    label = "Ünï-Çödé" * 9                      # 8 chars / 12 bytes each: 108 bytes per label
    body = ".".join([label] * 4)                # > 400 bytes, valid domain labels (high bytes are domain characters)
    key = body + ".example.com"
    b = M.DatabaseBuilder(build_epoch=7, case_insensitive=True)
    b.add_entry(key, {"long": 1})
    b.add_entry("short-ünï.example.org", {"short": 1})
    blob = b.build()
    b.close()
    # the public-suffix test of the extractor is case-sensitive (Q1): the suffix stays lower-case, the labels vary
    variants = [key, body.upper() + ".example.com", body.lower() + ".example.com",
                "".join(c.upper() if i % 3 else c.lower() for i, c in enumerate(body)) + ".example.com"]
    log = b"".join(b"GET http://" + v.encode() + b"/p " for v in variants) + "\nhost=SHORT-ÜNÏ.example.org x\n".encode() + b"http://" + key.encode()[:-5] + b"x.com/\n"
    gh, gl, gs, wh, wl, ws = _scan_both(M, oracle, blob, log)
    assert gs == ws
    assert gh == wh
    assert gl == wl
    assert len(wh) == 5
    db = M.Database(blob)
    odb = oracle.Database(blob)
    for q in variants + ["SHORT-ÜNÏ.example.org", key[:-1], key.upper()]:
        want, got = odb.lookup(q), db.lookup(q)
        if want["kind"] == "pattern" and want["data"] and want["data"][0] is not None:
            assert got == {"found": True, "prefix_len": 0, "data": want["data"][0]}, q
        else:
            assert got is None, (q, got)
    db.close()


def test_handmade_database_files(M, oracle):
    """The product reader, uploader and kernels on tests/golden/handmade_*.mxy — files assembled byte by byte from the format
    description (make_handmade_mxy.py), not by any builder of this repository: single queries give the answers that follow from
    the construction, and a scan of a log made of the query strings equals the oracle's."""
    exp = json.loads((GOLD / "handmade_expect.json").read_text())
    for name in ("24", "28", "32", "v6"):
        blob = (GOLD / f"handmade_{name}.mxy").read_bytes()
        db = M.Database(blob)
        md = db.metadata()
        assert md["record_size"] == exp[name]["record_size"] and md["ip_version"] == exp[name]["ip_version"]
        for q in exp[name]["queries"]:
            got, want = db.lookup(q["query"]), q["expect"]
            if want["kind"] == "ip":
                assert got == {"found": True, "prefix_len": want["prefix_len"], "data": want["data"]}, (name, q, got)
            elif want["kind"] == "pattern":
                assert got == {"found": True, "prefix_len": 0, "data": want["data"][0]}, (name, q, got)
            else:
                assert got is None, (name, q, got)
        db.close()
        log = b"".join(b"GET http://" + q["query"].encode() + b"/x HTTP/1.1\" \"ref=" + q["query"].encode() + b"\n" for q in exp[name]["queries"])
        gh, gl, gs, wh, wl, ws = _scan_both(M, oracle, blob, log)
        assert gs == ws
        assert gh == wh
        assert gl == wl
        assert len(wh) >= 30


def test_case_insensitive_special_casing_from_the_unicode_standard(M, oracle):
    """Lower-case forms hand-listed from the Unicode standard (UnicodeData / SpecialCasing: what Rust's str::to_lowercase
    implements), NOT taken from the table generator of this repository: a case-insensitive database keyed by the lower-case
    form must answer the upper-case spelling. Pins the device's mapping table and Final_Sigma rule independently of the
    tools/gen_lowercase.py / oracle pair, which share an origin."""
    pairs = [
        ("İstanbul.example.com", "i̇stanbul.example.com"),      # U+0130 -> U+0069 U+0307 (SpecialCasing)
        ("STRAẞE.example.com", "straße.example.com"),                 # U+1E9E -> U+00DF
        ("ΟΔΥΣΣΕΥΣ-1.example.com", "οδυσσευς-1.example.com"),         # Final_Sigma: preceded by a cased letter, not followed by one
        ("ΑΣ9.example.com", "ας9.example.com"),
        ("ΟΔΥΣΣΕΥΣ.example.com", "οδυσσευσ.example.com"),             # '.' is Case_Ignorable and 'e' is cased: NOT final here
        ("Σ.example.com", "σ.example.com"),                           # a lone capital sigma is not final
        ("Ǆungla.example.com", "ǆungla.example.com"),                 # U+01C4 -> U+01C6
        ("Kelvin.example.com", "kelvin.example.com"),            # KELVIN SIGN -> k
        ("Ångstrom.example.com", "ångstrom.example.com"),        # ANGSTROM SIGN -> U+00E5
        ("\U00010400\U00010401.example.com", "\U00010428\U00010429.example.com"),   # Deseret, 4-byte characters
        ("ÀÉÎÕÜ.example.com", "àéîõü.example.com"),
        ("ΆΈΉΊΌΎΏ.example.com", "άέήίόύώ.example.com"),
        ("ЖЁЛТЫЙ.example.com", "жёлтый.example.com"),
        # case pairs added after Unicode 13 (hand-listed from UnicodeData.txt of Unicode 14.0 / 16.0; the interpreter here knows neither)
        ("\u2C2Fx.example.com", "\u2C5Fx.example.com"),                 # 14.0 GLAGOLITIC CAUDATE CHU
        ("\uA7C0\uA7D0\uA7D6\uA7D8.example.com", "\uA7C1\uA7D1\uA7D7\uA7D9.example.com"),   # 14.0 Latin Extended-D
        ("\U00010570\U0001057A\U0001057C\U00010595.example.com", "\U00010597\U000105A1\U000105A3\U000105BC.example.com"),   # 14.0 Vithkuqi
        ("\u1C89a.example.com", "\u1C8Aa.example.com"),                 # 16.0 CYRILLIC TJE
        ("\uA7CBa\uA7DC.example.com", "\u0264a\u019B.example.com"),    # 16.0 RAMS HORN -> U+0264, LAMBDA WITH STROKE -> U+019B
        ("\uA7CC\uA7DA.example.com", "\uA7CD\uA7DB.example.com"),      # 16.0
        ("\U00010D50\U00010D65.example.com", "\U00010D70\U00010D85.example.com"),   # 16.0 Garay
    ]
    b = M.DatabaseBuilder(build_epoch=7, case_insensitive=True)
    for i, (_, low) in enumerate(pairs):
        b.add_entry(low, {"k": i})
    blob = b.build()
    b.close()
    db = M.Database(blob)
    odb = oracle.Database(blob)
    for i, (up, low) in enumerate(pairs):
        for q in (up, low):
            assert db.lookup(q) == {"found": True, "prefix_len": 0, "data": {"k": i}}, q
            assert odb.lookup(q)["data"] == [{"k": i}], q
    db.close()
    log = "".join(f"GET http://{up}/x http://{low}/y\n" for up, low in pairs).encode()
    gh, gl, gs, wh, wl, ws = _scan_both(M, oracle, blob, log)
    assert gs == ws and gh == wh and gl == wl and len(wh) == 2 * len(pairs)


def test_ac_node_walk_without_the_flattened_automaton(M, oracle, monkeypatch):
    """MATCHY_AMD_DFA_MAX_MB=0 at open: the database is uploaded WITHOUT the dense DFA, the AC prefilters are off and
    glob_find_all walks the stored ACNodeHot records — goto over ONE / SPARSE / DENSE nodes, failure links until the root
    (paraglob_offset.rs:1186-1353) — which is the path of a database whose flattened automaton does not fit (10 M globs).
    Scans (both entries, sliced too) and single queries against the oracle: glob fuzz cases, a case-insensitive database
    (text ASCII-folded while it is walked), the handmade file with all four node kinds, results beyond the lane storage."""
    monkeypatch.setenv("MATCHY_AMD_DFA_MAX_MB", "0")
    for seed in (11, 14):
        pats, log = _glob_fuzz_case(seed)
        b = M.DatabaseBuilder(build_epoch=6)
        for pt, i in pats.items():
            b.add_entry(pt, {"g": i})
        b.add_entry("literal:" + log.split()[0].decode(), {"lit": True})
        blob = b.build()
        db = M.Database(blob)
        assert M.lib().matchy_amd_ac_dfa_states(db.handle) == 0    # the node walk really is what runs
        db.close()
        gh, gl, gs, wh, wl, ws = _scan_both(M, oracle, blob, log)
        assert gs == ws and gh == wh and gl == wl
        assert len(gh) > 100
    entries, log, names = _ci_case(23)
    b = M.DatabaseBuilder(build_epoch=7, case_insensitive=True)
    for k, v in entries:
        b.add_entry(k, v)
    blob = b.build()
    b.close()
    gh, gl, gs, wh, wl, ws = _scan_both(M, oracle, blob, log)
    assert gs == ws and gh == wh and gl == wl
    assert len(gh) > 60
    db = M.Database(blob)
    assert M.lib().matchy_amd_ac_dfa_states(db.handle) == 0
    odb = oracle.Database(blob)
    rng = random.Random(23)
    for nm in names[:150]:
        q = "".join(ch.upper() if rng.random() < 0.5 else ch.lower() for ch in nm)
        want, got = odb.lookup(q), db.lookup(q)
        if want["kind"] == "pattern" and want["data"] and want["data"][0] is not None:
            assert got == {"found": True, "prefix_len": 0, "data": want["data"][0]}, q
        elif want["kind"] != "pattern":
            assert got is None, (q, got)
    db.close()
    # handmade automaton (EMPTY / ONE / SPARSE / DENSE nodes, built by the textbook algorithm, not by this repository's builder)
    exp = json.loads((GOLD / "handmade_expect.json").read_text())
    for name in ("24", "v6"):
        blob = (GOLD / f"handmade_{name}.mxy").read_bytes()
        db = M.Database(blob)
        assert M.lib().matchy_amd_ac_dfa_states(db.handle) == 0
        for q in exp[name]["queries"]:
            got, want = db.lookup(q["query"]), q["expect"]
            if want["kind"] == "ip":
                assert got == {"found": True, "prefix_len": want["prefix_len"], "data": want["data"]}, (name, q, got)
            elif want["kind"] == "pattern":
                assert got == {"found": True, "prefix_len": 0, "data": want["data"][0]}, (name, q, got)
            else:
                assert got is None, (name, q, got)
        db.close()
        log = b"".join(b"GET http://" + q["query"].encode() + b"/x HTTP/1.1\" \"ref=" + q["query"].encode() + b"\n" for q in exp[name]["queries"])
        gh, gl, gs, wh, wl, ws = _scan_both(M, oracle, blob, log)
        assert gs == ws and gh == wh and gl == wl
    # with the automaton back (default limit) the same database reports its states
    monkeypatch.delenv("MATCHY_AMD_DFA_MAX_MB")
    db = M.Database((GOLD / "handmade_24.mxy").read_bytes())
    assert M.lib().matchy_amd_ac_dfa_states(db.handle) > 0
    db.close()


def test_wide_label_start_class_with_another_public_suffix_list(M, tmp_path):
    """k_anchor<false> with tl_wide: the shipped public-suffix list only has last labels that start with a-z or a byte >= 0x80, so
    the streaming pass uses that narrow first-byte class; a list with a last label that starts with a digit (MATCHY_AMD_PSL)
    switches it to "any label byte or '-'" (k_anchor.hip anchor_tl_wide). The list is loaded once per process, so the comparison
    with the oracle (same list) runs in a child process: extractor fuzz + a scan."""
    import subprocess
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
    from tools import gen_psl
    base = Path(__file__).resolve().parent.parent / "matchy_amd" / "data" / "psl.bin"
    # decode the shipped container and add suffixes whose last label starts with a digit / is all digits
    raw = base.read_bytes()
    count = int.from_bytes(raw[8:12], "little")
    sufs, prev, p = [], b"", 16
    for _ in range(count):
        shared, rest = raw[p], raw[p + 1]
        s = prev[:shared] + raw[p + 2:p + 2 + rest]
        p += 2 + rest
        sufs.append(s)
        prev = s
    sufs = sorted(set(sufs + [b"4u", b"co.4u", b"7z", b"123", b"0x0"]))
    pslp = tmp_path / "psl_wide.bin"
    pslp.write_bytes(gen_psl.encode(sufs))
    code = r"""
import random, sys
sys.path.insert(0, ".")
import matchy_amd as M
from oracle import oracle
from tools import synth
ex = M.Extractor()
rng = random.Random(77)
cases = [b"shop.4u", b"a.co.4u x", b"file.7z\n", b"v1.2.123 ", b"x.0x0", b"host.4u.", b"1.2.3.4", b"a.4u-b", b"www.shop.4u/p", b"q.7zz", b"9.123", b"a.b.7z c.d.e.4u"]
alphas = [b"ab.47uz0x-", b"0123456789.", b"a1.:@ /-\n7z4u", b"comnetorg.uk.co.4u7z"]
for it in range(300):
    alpha = rng.choice(alphas)
    n = rng.choice([5, 17, 64, 129, 1000, 1025, 5000])
    cases.append(bytes(rng.choice(alpha) for _ in range(n)))
seen = 0
for buf in cases:
    got = [(t, s, e, v) for (t, s, e, v) in ex.extract_from_chunk(buf)]
    want = [(t, s, e, v) for (t, s, e, v) in oracle.extract(buf)]
    assert got == want, buf
    seen += sum(1 for (t, s, e, v) in got if t == "Domain" and v.rsplit(".", 1)[-1][:1].isdigit())
assert seen > 10, seen
cfg = synth.config("c2/50")
blob = synth.build_db(cfg)
log = synth.make_log(cfg, 0, 3000) + b"GET http://shop.4u/ x.co.4u y.7z 8.8.8.8.123\n" * 50
db = M.Database(blob)
sc = M.Scanner(db)
r = sc.scan(log)
hits, stats = r.hits(), (r.lines, r.candidates)
r.close()
want, _, st = oracle.Database(blob).scan(log, want_json=False)
assert stats == (st.lines, st.candidates), (stats, st.lines, st.candidates)
assert hits == want
print("WIDE-OK", seen, len(hits))
"""
    root = Path(__file__).resolve().parent.parent
    env = dict(os.environ, MATCHY_AMD_PSL=str(pslp))
    out = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    assert "WIDE-OK" in out.stdout


@pytest.mark.parametrize("ci", [False, True])
@pytest.mark.parametrize("first", [".", "-"])
def test_suffix_globs_take_the_suffix_filter(M, oracle, ci, first):
    """Databases whose globs are all *LITERAL with one common first byte (what "*.domain" indicator lists are): k_validate_dom
    decides from a few hashed suffixes of the name whether a glob can match (DevDb::sfx_bm) instead of walking the automaton, and
    tells the lookup pass (Candidate::pad). Names that end with a literal, contain one in the middle, equal it without the leading
    byte, differ in case; literals with one, two and three occurrences of the first byte; a literal of 33 bytes (cannot end a short
    name) and one of 2 bytes (never matches, Q8); literal keys beside the globs. Scan (both entries) and queries vs the oracle."""
    rng = random.Random(99 + ci)
    f = first
    lits = [f"{f}evil{i}.com" for i in range(40)] + [f"{f}a{f}bad{i}.net" for i in range(20)] + [f"{f}x{f}y{f}deep{i}.org" for i in range(10)] + \
           [f"{f}co", f"{f}" + "long" * 8 + ".com", f"{f}Mixed{f}Case.io"]
    b = M.DatabaseBuilder(build_epoch=9, case_insensitive=ci)
    for i, l in enumerate(lits):
        b.add_entry("*" + l, {"g": i})
    b.add_entry("plain-key.example.com", {"lit": 1})
    b.add_entry("evil3.com", {"lit": 2})            # a literal key that is also the tail of a glob literal
    blob = b.build()
    b.close()
    names = []
    for l in lits:
        tail = l[1:] if f == "." else l
        for pre in ("www", "a.b", "x-y", "WWW", ""):
            names.append(pre + l)                    # ends with the literal
        names.append(tail)                           # the literal without its first byte: no match for "." literals
        names.append("www" + l + ".attacker.net")    # literal in the middle
        names.append("www" + l.upper())
        names.append("www" + l[:-1] + "x")
    names += ["plain-key.example.com", "evil3.com", "host.evil3.com", "evil3.com.evil4.com", "a.b.c.d.e.f.g.evil5.com", "nothing.example.org"]
    rng.shuffle(names)
    log = b"".join(rng.choice([b"GET http://", b"host=", b" "]) + nm.encode() + rng.choice([b"/x ", b" ", b"\n", b"\" "]) for nm in names) + b"\n"
    gh, gl, gs, wh, wl, ws = _scan_both(M, oracle, blob, log)
    assert gs == ws
    assert gh == wh
    assert gl == wl
    assert len(gh) > 150
    db = M.Database(blob)
    assert M.lib().matchy_amd_suffix_filter(db.handle) == 1      # the suffix filter really is what ran
    odb = oracle.Database(blob)
    for nm in names:
        want, got = odb.lookup(nm), db.lookup(nm)
        if want["kind"] == "pattern" and want["data"] and want["data"][0] is not None:
            assert got == {"found": True, "prefix_len": 0, "data": want["data"][0]}, nm
        elif want["kind"] != "pattern":
            assert got is None, (nm, got)
    db.close()
    # one glob of another shape (prefix) and the database takes the automaton walk again
    b = M.DatabaseBuilder(build_epoch=9, case_insensitive=ci)
    for i, l in enumerate(lits[:5]):
        b.add_entry("*" + l, {"g": i})
    b.add_entry("evil*", {"p": 1})
    db = M.Database(b.build())
    b.close()
    assert M.lib().matchy_amd_suffix_filter(db.handle) == 0
    db.close()


def _xmr_kat():
    return json.loads((GOLD / "xmr_kat.json").read_text())


def test_monero_accept_branch_against_constructed_vectors(M, gpu_extractor, oracle):
    """E9 (lib.rs:1367-1409, 1895-1920): checksum-valid Monero tokens — constructed by tests/golden/make_xmr_kat.py from the rule,
    not taken from the oracle — through matchy_extractor_extract_chunk: GPU == oracle == the construction; every accept with one
    flipped character and with its last character flipped is a reject; accepts straddling the row (64 B), block (2 KiB) and
    segment edges of the streaming pass, at the very end of the buffer, and two in a row."""
    k = _xmr_kat()
    for a in k["accept"]:
        ab = a.encode()
        for pre, post in ((b"pay ", b" now"), (b"", b""), (b"addr=", b"\n"), (b"[", b"]"), (b"\n", b"\n")):
            buf = pre + ab + post
            got = norm(gpu_extractor.extract_from_chunk(buf))
            assert got == norm(oracle.extract(buf)), buf
            assert [m for m in got if m[0] == "Monero"] == [("Monero", len(pre), len(pre) + len(a), a)], buf
        buf = b"x" + ab
        got = norm(gpu_extractor.extract_from_chunk(buf))
        assert got == norm(oracle.extract(buf)) and not [m for m in got if m[0] == "Monero"]
    for r in k["reject"]:
        buf = b"pay " + r["text"].encode() + b" now\n"
        got = norm(gpu_extractor.extract_from_chunk(buf))
        assert got == norm(oracle.extract(buf)), r
        assert not [m for m in got if m[0] == "Monero"], r
    picks = [a for a in k["accept"] if len(a) in (95, 106)][:4] + [k["accept"][-1]]
    for edge in (64, 256, 2048, 4096, 8192, 16384, 32768, 65536):
        for a in picks:
            ab = a.encode()
            for shift in (0, 1, 2, 3, 4, 5, 31, 32, 33, 63, 64, 65, len(ab) - 5, len(ab) - 4, len(ab) - 1, len(ab), len(ab) + 1):
                pre = edge - shift
                if pre < 1:
                    continue
                buf = b"a" * (pre - 1) + b" " + ab + b" tail\n"
                got = norm(gpu_extractor.extract_from_chunk(buf))
                assert got == norm(oracle.extract(buf)), (edge, a, shift)
                assert ("Monero", pre, pre + len(ab), a) in got, (edge, a, shift)
    for a in picks:
        for pad in (0, 1, 15, 63, 64, 2047, 2048 - len(a)):
            buf = b"x" * pad + b" " + a.encode()
            got = norm(gpu_extractor.extract_from_chunk(buf))
            assert got == norm(oracle.extract(buf)) and ("Monero", pad + 1, pad + 1 + len(a), a) in got, (a, pad)
    buf = " ".join(k["accept"]).encode() + b"\n" + ",".join(r["text"] for r in k["reject"]).encode() + b"\n"
    got = norm(gpu_extractor.extract_from_chunk(buf))
    assert got == norm(oracle.extract(buf))
    assert [m[3] for m in got if m[0] == "Monero"] == k["accept"]


def test_monero_hits_through_scans_and_a_hash_dense_batch(M, oracle):
    """The same vectors as DATABASE KEYS and log tokens: a small log through every scan entry (host buffer, device-resident
    forked / sliced / submitted / compact), then spliced into a HASH-DENSE batch (k_validate<4> beside k_rare, > 256 K long-token
    anchors) of the c2/10 database extended by the addresses. Every accept that is a key must be a hit, no reject may be."""
    from tools import synth
    k = _xmr_kat()
    keys = k["accept"][::2]
    b = M.DatabaseBuilder(build_epoch=1)
    for i, a in enumerate(keys):
        b.add_entry(a, {"coin": "xmr", "n": i})
    b.add_entry("evil.com", {"why": "bad"})
    b.add_entry("8.8.8.8", {"who": "dns"})
    blob = b.build()
    rows = [f"203.0.113.{i} paid {a} via evil.com" for i, a in enumerate(k["accept"])]
    rows += [f"8.8.8.8 refused {r['text']}" for r in k["reject"]]
    text = ("\n".join(rows) + "\n").encode()
    gh, gl, gs, wh, wl, ws = _scan_both(M, oracle, blob, text)
    assert gs == ws and gh == wh and gl == wl
    got_xmr = [text[h["start"]:h["end"]].decode() for h in gh if h["type"] == "Monero"]
    assert got_xmr == keys

    cfg = synth.config("c2/10")
    b = M.DatabaseBuilder(build_epoch=1)
    for key, data in synth.ioc_entries(cfg):
        b.add_entry(key.decode(), json.loads(data))
    for i, a in enumerate(keys):
        b.add_entry(a, {"coin": "xmr", "n": i})
    blob = b.build()
    dense = synth.make_log(cfg, 3_000_000, 300000, shape="hash-dense")
    lines = dense.split(b"\n")
    everything = k["accept"] + [r["text"] for r in k["reject"]]
    rng = random.Random(9)
    for j, tok in enumerate(everything * 40):
        at = rng.randrange(len(lines) - 1)
        lines[at] = lines[at] + (b" xmr=" if j & 1 else b" ") + tok.encode()
    dense = b"\n".join(lines)
    db = M.Database(blob); sc = M.Scanner(db)
    odb = oracle.Database(blob)
    want, _, st = odb.scan(dense, threads=min(len(os.sched_getaffinity(0)), 16), cache=0, want_json=False)
    n_xmr = sum(1 for h in want if h["type"] == "Monero")
    assert n_xmr == 40 * len(keys) and st.candidates > 600000
    res = sc.scan(dense)
    assert (res.lines, res.candidates) == (st.lines, st.candidates)
    assert res.hits() == want
    res.close()
    _device_entries(sc, dense, want, None, (st.lines, st.candidates), slices=(3,))
    sc.close(); db.close()


def test_tree_record_vectors_of_the_reference(M, oracle):
    """mmdb/tree.rs:322-398 as whole files (tests/golden/make_tree_kat.py): node 0 holds exactly the bytes the reference's tests
    write — 24-bit `000001 000002`, 28-bit `000001 12 000002` (records 0x1000001 / 0x2000002: non-zero high nibbles, data
    pointers 16 and 32 MiB into the data section), the same two as 32-bit words, and test_calculate_data_offset's node_count 100 /
    records 116 and 200. Single queries give the answers that follow from the construction; a scan equals the oracle's."""
    from tests.test_builder_oracle import _tree_kat
    for name, (blob, node0, queries) in _tree_kat().items():
        db = M.Database(blob)
        for q, want in queries:
            got = db.lookup(q)
            if want is None:
                assert got is None, (name, q, got)
            else:
                assert got == {"found": True, "prefix_len": want[0], "data": want[1]}, (name, q, got)
        db.close()
        log = b"".join(b"%s - - [x] \"GET /a?from=%s HTTP/1.1\"\n" % (q.encode(), q.encode()) for q, _ in queries)
        gh, gl, gs, wh, wl, ws = _scan_both(M, oracle, blob, log, compact_fits=len(blob) < (1 << 22))
        assert gs == ws and gh == wh and gl == wl
        assert len(gh) == 2 * sum(1 for _, w in queries if w is not None)


def test_multi_scanner_back_pressure(M, oracle):
    """matchy_multi_scanner_submit blocks once max_pending() batches are out (queued, being scanned, or finished and not taken) —
    the reference's bounded channels (processing/parallel.rs:563-577) — so a slow gatherer cannot make the reader buffer the whole
    input. A thread that only submits stalls at the bound until the main thread takes results; the results come back in
    submission order and equal the single scanner's."""
    import ctypes
    import threading
    import time
    from tools import synth
    cfg = synth.config("c2/20")
    blob = synth.build_db(cfg)
    db = M.Database(blob)
    ms = M.MultiScanner(db, devices=(0, 0))
    limit = ms.max_pending()
    assert limit == 2 * 2 + 2
    logs = [synth.make_log(cfg, 1000 * i, 1000) for i in range(3 * limit)]
    bufs = [ctypes.create_string_buffer(l, len(l)) for l in logs]
    done_submitting = threading.Event()

    def feeder():
        for i, b in enumerate(bufs):
            ms.submit_ptr(ctypes.addressof(b), len(logs[i]), tag=i)
        done_submitting.set()

    t = threading.Thread(target=feeder)
    t.start()
    time.sleep(1.0)                       # nobody gathers: the feeder must be stuck at the bound, not through its list
    assert not done_submitting.is_set()
    assert ms.pending() == limit
    sc = M.Scanner(db)
    for i, l in enumerate(logs):
        b = ms.next(want_hits=True)
        assert ms.pending() <= limit
        assert (b["seq"], b["tag"]) == (i, i)
        r = sc.scan(l)
        assert (b["lines"], b["candidates"], b["hits"]) == (r.lines, r.candidates, r.hits())
        r.close()
    t.join()
    assert done_submitting.is_set() and ms.pending() == 0 and ms.next() is None
    assert all(n >= -1 for n, _ in ms.worker_numa())
    sc.close(); ms.close(); db.close()


def _query_probe_set(seed, cfgname):
    """(database blob, queries): keys of the configuration, names / addresses cut from its log, near misses"""
    from tools import synth
    cfg = synth.config(cfgname)
    blob = synth.build_db(cfg)
    rng = random.Random(seed)
    keys = [k.decode() for k, _ in synth.ioc_entries(cfg)]
    qs = rng.sample(keys, min(len(keys), 400))
    for k in list(qs):
        if "/" in k and ":" not in k:
            qs.append(k.split("/")[0])                      # an address inside a CIDR key
        if k.startswith("*."):
            qs += ["www" + k[1:], "a.b" + k[1:], k[2:]]     # names under a suffix glob, and the bare suffix
        if k.startswith("glob:"):
            qs += [k[5:], "x" + k[5:] + "y"]                # substring semantics of literal patterns (Q9)
    log = synth.make_log(cfg, 0, 3000)
    import re
    toks = re.findall(rb"[0-9A-Za-z][0-9A-Za-z.:\-]{3,80}", log)
    qs += [t.decode() for t in rng.sample(toks, 600)]
    qs += ["", ".", "1.2.3.4", "255.255.255.255", "0.0.0.0", "::", "::1", "2001:db8::1", "::ffff:1.2.3.4", "1.2.3", "münchen.de", "x" * 300,
           "EXAMPLE.COM", "a" * 70000]
    return blob, qs


@pytest.mark.parametrize("cfgname", ["c1", "c2/20", "c3b/50", "c4/20", "c5/100"])
def test_host_query_path_against_oracle_and_kernels(M, oracle, cfgname):
    """matchy_query / matchy_amd_query_json answer on the HOST since round 5 (csrc/host_lookup.cpp: SURVEY §8b "single queries stay
    on the CPU path"; trie walk, literal probe, Paraglob::find_all written against the on-disk sections — not the oracle). Every
    probe against the oracle's Database::lookup: same verdict, same prefix length, same data values in the same order; and the same
    probes through the lookup KERNELS (MATCHY_AMD_QUERY_ON_GPU=1 in a child process) give byte-identical JSON."""
    import subprocess
    import sys
    import tempfile
    blob, qs = _query_probe_set(5, cfgname)
    db = M.Database(blob)
    odb = oracle.Database(blob)
    host_answers = []
    n_found = 0
    for q in qs:
        want = odb.lookup(q)
        found, arr = db.query_json(q)
        host_answers.append(json.loads(json.dumps([found, arr])))   # (a copy: the checks below take the IP answer apart)
        got1 = db.lookup(q)
        if want["kind"] == "ip":
            n_found += 1
            assert found and len(arr) == 1 and arr[0].pop("prefix_len") == want["prefix_len"] and arr[0].pop("cidr"), (q, arr)
            assert arr[0] == want["data"], (q, arr, want)
            assert got1 == {"found": True, "prefix_len": want["prefix_len"], "data": want["data"]}, q
        elif want["kind"] == "pattern":
            n_found += 1
            assert found and arr == [d for d in want["data"] if d is not None], (q, arr, want)
            first = next((d for d in want["data"] if d is not None), None)
            # matchy_query: the FIRST pattern's data only (c_api/matchy.rs:1143-1154)
            if want["data"] and want["data"][0] is not None:
                assert got1 == {"found": True, "prefix_len": 0, "data": want["data"][0]}, q
            elif first is None:
                assert got1 is None, q
        else:
            assert not found and arr == [] and got1 is None, (q, arr)
    assert n_found > 100
    db.close()
    with tempfile.TemporaryDirectory() as td:
        (Path(td) / "db.mxy").write_bytes(blob)
        (Path(td) / "q.json").write_text(json.dumps(qs))
        code = r"""
import json, sys
sys.path.insert(0, %r)
import matchy_amd as M
db = M.Database(open(sys.argv[1], "rb").read())
out = []
for q in json.load(open(sys.argv[2])):
    f, a = db.query_json(q)
    out.append([f, a])
json.dump(out, open(sys.argv[3], "w"))
""" % str(ROOT)
        env = dict(os.environ, MATCHY_AMD_QUERY_ON_GPU="1")
        p = subprocess.run([sys.executable, "-c", code, str(Path(td) / "db.mxy"), str(Path(td) / "q.json"), str(Path(td) / "out.json")],
                           env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stderr[-2000:]
        gpu_answers = json.loads((Path(td) / "out.json").read_text())
    for q, h, g in zip(qs, host_answers, gpu_answers):
        assert h == g, q


def test_glob_vectors_of_the_reference_through_matchy_query(M):
    """glob.rs:464-705 through `matchy_query`: on the host path (the default) and, in a child process with MATCHY_AMD_QUERY_ON_GPU=1, through the
    lookup kernels (DFA walk, glob matcher with its 100 000-step budget, UTF-8 stepping, case folding)."""
    import subprocess
    import sys
    import tempfile
    from tests.test_builder_oracle import _glob_kat_dbs
    dbs = _glob_kat_dbs()
    for c, blob in dbs:
        db = M.Database(blob)
        for t in c["match"]:
            assert db.lookup(t) == {"found": True, "prefix_len": 0, "data": {"p": 1}}, (c["ref"], c["pattern"], t)
        for t in c["nomatch"]:
            assert db.lookup(t) is None, (c["ref"], c["pattern"], t)
        db.close()
    with tempfile.TemporaryDirectory() as td:
        spec = []
        for i, (c, blob) in enumerate(dbs):
            (Path(td) / f"{i}.mxy").write_bytes(blob)
            spec.append({"db": str(Path(td) / f"{i}.mxy"), "match": c["match"], "nomatch": c["nomatch"], "ref": c["ref"]})
        (Path(td) / "spec.json").write_text(json.dumps(spec))
        code = r"""
import json, sys
sys.path.insert(0, %r)
import matchy_amd as M
for s in json.load(open(sys.argv[1])):
    db = M.Database(open(s["db"], "rb").read())
    for t in s["match"]:
        assert db.lookup(t) is not None, (s["ref"], t)
    for t in s["nomatch"]:
        assert db.lookup(t) is None, (s["ref"], t)
    db.close()
print("OK")
""" % str(ROOT)
        p = subprocess.run([sys.executable, "-c", code, str(Path(td) / "spec.json")], env=dict(os.environ, MATCHY_AMD_QUERY_ON_GPU="1"),
                           capture_output=True, text=True, timeout=600)
        assert p.returncode == 0 and "OK" in p.stdout, p.stderr[-2000:]


def test_query_tests_again_through_the_lookup_kernels():
    """Since round 5 `matchy_query` answers on the host; the tests that pin the LOOKUP KERNELS through single queries (handmade files,
    the reference's behaviour vectors, tree record vectors, Unicode special casing, the structured-data walkers) run a second time in a
    child process with MATCHY_AMD_QUERY_ON_GPU=1, so that both paths stay pinned by the same vectors."""
    import subprocess
    import sys
    sel = ("handmade_database_files or reference_behaviour_vectors or single_query_api or tree_record_vectors or special_casing "
           "or structured_data_walkers or case_insensitive_long_non_ascii_key or paraglob_integration_vectors or literal_hash_vectors or per_handle")
    env = dict(os.environ, MATCHY_AMD_QUERY_ON_GPU="1")
    p = subprocess.run([sys.executable, "-m", "pytest", str(Path(__file__)), "-q", "-x", "-m", "gpu", "-k", sel, "-p", "no:cacheprovider"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=str(ROOT))
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-2000:]
    assert " passed" in p.stdout and "failed" not in p.stdout, p.stdout[-1000:]


def test_query_cache_is_per_handle_and_handles_are_isolated(M, tmp_path):
    """crates/matchy/tests/cache_stale_notfound_test.rs (a cached NotFound — or a cached match — of one database must not answer for the
    next one opened, same path rewritten included) and sequential_builder_test.rs (databases built one after the other, queried
    interleaved, each answer only its own keys), through matchy_open / matchy_query with the query cache on. The cache here is per thread
    AND handle (capi.cpp Db::cache): a new handle starts empty whatever address or path it has."""
    import threading

    def build(entries):
        b = M.DatabaseBuilder(build_epoch=7)
        for k, v in entries:
            b.add_entry(k, v)
        blob = b.build()
        b.close()
        return blob
    ip = "80.239.174.89"
    a, bb = build([("1.1.1.0/24", {"name": "other"})]), build([("80.239.0.0/16", {"name": "matched"})])
    # cache_stale_notfound_test.rs:37-135 — NotFound cached by A must not persist into B ...
    for first, second, want in ((a, bb, {"found": True, "prefix_len": 16, "data": {"name": "matched"}}), (bb, a, None)):   # ... nor a match (:138-222)
        d1 = M.Database(first)
        for _ in range(3):
            d1.lookup(ip)
        d1.close()
        d2 = M.Database(second)
        assert d2.lookup(ip) == want and d2.lookup(ip) == want
        d2.close()
    # :389-460 — the same PATH rewritten and reopened
    path = tmp_path / "db.mxy"
    path.write_bytes(a)
    d = M.Database(str(path))
    assert d.lookup(ip) is None
    d.close()
    os.chmod(path, 0o644)
    path.write_bytes(bb)
    d = M.Database(str(path))
    assert d.lookup(ip) == {"found": True, "prefix_len": 16, "data": {"name": "matched"}}
    d.close()
    # sequential_builder_test.rs:23-100, 210-285 — three databases open at once, interleaved queries, also from other threads
    dbs = [M.Database(build([(f"{i}.{i}.{i}.{i}", {"db": i}), (f"host{i}.example.com", {"db": i}), (f"*.zone{i}.test", {"db": i})])) for i in (1, 2, 3)]

    def rounds(n, errs):
        try:
            for r in range(n):
                for i, d in zip((1, 2, 3), dbs):
                    for j in (1, 2, 3):
                        for q, hit in ((f"{j}.{j}.{j}.{j}", {"found": True, "prefix_len": 32, "data": {"db": j}}),
                                       (f"host{j}.example.com", {"found": True, "prefix_len": 0, "data": {"db": j}}),
                                       (f"a.zone{j}.test", {"found": True, "prefix_len": 0, "data": {"db": j}})):
                            got = d.lookup(q)
                            assert got == (hit if i == j else None), (r, i, j, q, got)
        except AssertionError as e:
            errs.append(e)
    errs = []
    rounds(10, errs)
    ths = [threading.Thread(target=rounds, args=(5, errs)) for _ in range(4)]
    [t.start() for t in ths]; [t.join() for t in ths]
    assert not errs, errs[:1]
    st = dbs[0].stats()
    assert st["cache_hits"] > 0 and st["total_queries"] == st["cache_hits"] + st["cache_misses"]
    for d in dbs:
        d.close()


def test_paraglob_integration_vectors_through_matchy_query(M):
    """matchy-paraglob/tests/integration_tests.rs:11-260 (tests/test_builder_oracle.py PARAGLOB_KAT) through matchy_amd_query_json: the ids of the
    patterns that matched come back as their data values. Runs on the host path here and through the lookup kernels in the child process of
    test_query_tests_again_through_the_lookup_kernels."""
    from tests.test_builder_oracle import PARAGLOB_KAT, paraglob_kat_check, paraglob_kat_db
    for ref, ci, patterns, checks in PARAGLOB_KAT:
        db = M.Database(paraglob_kat_db(patterns, ci))
        for text, expect in checks:
            found, arr = db.query_json(text)
            ids = [d["i"] for d in arr]
            assert paraglob_kat_check(ids, expect) and found == bool(ids), (ref, text, expect, ids)
        db.close()


def test_literal_hash_vectors_through_matchy_query(M):
    """crates/matchy/tests/test_literal_hash.rs:52-265 (tests/test_builder_oracle.py LITERAL_HASH_KAT) through matchy_amd_query_json and
    matchy_query: literal and glob on one text give two data objects, the literal's first; literals holding glob characters match exactly and
    nothing else. Host path here, lookup kernels in the child process of test_query_tests_again_through_the_lookup_kernels."""
    from tests.test_builder_oracle import LITERAL_HASH_KAT, build
    for ref, entries, checks in LITERAL_HASH_KAT:
        db = M.Database(build(entries))
        for q, kind, n in checks:
            found, arr = db.query_json(q)
            r = db.lookup(q)
            if kind == "notfound":
                assert not found and arr == [] and r is None, (ref, q, arr)
            elif kind == "ip":
                assert found and len(arr) == 1 and arr[0]["prefix_len"] == 32 and r["found"] and r["prefix_len"] == 32, (ref, q, arr)
            else:
                assert found and len(arr) == n and r["found"] and r["prefix_len"] == 0 and r["data"] == arr[0], (ref, q, arr, r)
        if ref.endswith(":53"):
            assert [d["source"] for d in db.query_json("evil.com")[1]] == ["literal", "glob"]
        db.close()


@pytest.mark.parametrize("alnum_literal", [False, True])
def test_long_tokens_in_databases_with_globs(M, oracle, alnum_literal):
    """DevDb::ac_alnum: when no literal of the glob automaton consists of letters and digits only, a long token (hex hash, Base58 / 0x address)
    cannot contain one — the scan then decides tokens through the literal table alone (bitmap in k_validate<4>, no automaton walk, no glob
    pass). With one such literal in the database (`glob:deadbeef`, `*cafe12*`) the tokens take the glob pass as before. Both ways: GPU == oracle,
    through every device entry, over tokens that are literal keys, that contain the substring literal, and that are neither."""
    import hashlib
    rng = random.Random(77)
    keys = [hashlib.md5(b"k%d" % i).hexdigest() for i in range(40)] + [hashlib.sha256(b"k%d" % i).hexdigest() for i in range(40)] + \
           [hashlib.sha1(b"k%d" % i).hexdigest() for i in range(20)]
    b = M.DatabaseBuilder(build_epoch=3)
    for i, k in enumerate(keys):
        b.add_entry(k, {"h": i})
    for k, v in [("*.evil-domain.com", {"g": 1}), ("glob:phish.example", {"g": 2}), ("bad-??.example.[a-c]om", {"g": 3}), ("203.0.113.0/24", {"n": 1}),
                 ("mail.corp.example", {"l": 1})]:
        b.add_entry(k, v)
    if alnum_literal:
        b.add_entry("glob:deadbeef", {"g": "substring"})
        b.add_entry("*cafe12*", {"g": "star"})
    blob = b.build()
    lines = []
    for i in range(6000):
        r = rng.random()
        if r < 0.2:
            tok = rng.choice(keys)
        elif r < 0.4:
            h = hashlib.sha256(b"x%d" % i).hexdigest()
            tok = h[:20] + rng.choice(["deadbeef", "cafe1234"]) + h[28:]
        elif r < 0.5:
            tok = "0x" + hashlib.sha1(b"e%d" % i).hexdigest()                # Ethereum-shaped (all lower case: accepted)
        elif r < 0.6:
            tok = "1A1zP1eP5QGefi2DMPTfTL5SLmv7DivfNa"                       # a valid Base58Check address
        else:
            tok = hashlib.new(rng.choice(["md5", "sha1", "sha256", "sha512"]), b"r%d" % i).hexdigest()
        host = rng.choice(["www.evil-domain.com", "phish.example.net", "mail.corp.example", "bad-ab.example.com", "ok.example.org"])
        lines.append(f"203.0.113.{i % 250} {host} id={tok} user=a{i}@{host}".encode())
    text = b"\n".join(lines) + b"\n"
    gh, gl, gs, wh, wl, ws = _scan_both(M, oracle, blob, text)
    assert gs == ws and gh == wh and gl == wl
    assert len(gh) > 3000, len(gh)


@pytest.mark.parametrize("ci", [False, True])
def test_email_addresses_as_database_keys(M, oracle, ci):
    """E-mail candidates go through the literal-key bitmap before they are listed (lit_bm_may_hit: first 32 bytes + length, ASCII-folded for a
    case-insensitive database): a database keyed with a sample of the addresses of a generated log — short ones, ones longer than 32 bytes,
    upper-case variants of the keys in the log — must give the oracle's match set through every device entry."""
    buf = _long_emails(5, count=2500)
    found = sorted({v for t, s, e, v in oracle.extract(buf) if t == "Email"})
    assert len(found) > 300 and any(len(v) > 32 for v in found) and any(len(v) <= 32 for v in found)
    keys = found[::3]
    b = M.DatabaseBuilder(build_epoch=4, case_insensitive=ci)
    for v in keys:
        b.add_entry("literal:" + v, {"n": len(v)})
    b.add_entry("192.0.2.1", {"ip": True})
    blob = b.build()
    # the same addresses again in upper case (hits only in the case-insensitive database) and with one byte changed (never hits)
    extra = bytearray()
    for v in keys[:200]:
        extra += b" " + v.upper().encode() + b" " + (v[:-1] + ("x" if v[-1] != "x" else "y")).encode() + b"\n"
    text = buf + b"\n" + bytes(extra)
    gh, gl, gs, wh, wl, ws = _scan_both(M, oracle, blob, text)
    assert gs == ws and gh == wh and gl == wl
    assert len(gh) >= len(keys)
