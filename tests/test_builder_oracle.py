"""CPU: databases written by our builder (host C++, matchy_builder_* C ABI) are read by the oracle — an independent
restatement of the reference reader — and must give the answers the reference's own tests assert
(crates/matchy/tests/test_ip_longest_prefix_match.rs, test_ip_exact_match.rs, test_literal_hash.rs,
crates/matchy-paraglob/tests/integration_tests.rs, crates/matchy-paraglob/src/paraglob_offset.rs:1890-1944).
No GPU calls: only the builder half of the library is exercised here."""
import json
from pathlib import Path

import pytest

import matchy_amd as M


def build(entries, epoch=1700000000):
    b = M.DatabaseBuilder(build_epoch=epoch)
    for k, v in entries:
        b.add_entry(k, v)
    blob = b.build()
    b.close()
    return blob


def test_library_exports_every_declared_symbol():
    L = M.lib()
    for name in M.EXPORTED_SYMBOLS:
        assert hasattr(L, name), name
    # header and list agree
    import re
    from pathlib import Path
    hdr = (Path(__file__).parent.parent / "include" / "matchy_amd.h").read_text()
    declared = set(re.findall(r"\b(matchy_[a-z0-9_]+)\s*\(", hdr))
    declared = {d for d in declared if not d.endswith("_t")}
    declared -= set(re.findall(r"static inline \w+ (matchy_[a-z0-9_]+)\s*\(", hdr))   # header-only helpers
    assert declared == set(M.EXPORTED_SYMBOLS), declared ^ set(M.EXPORTED_SYMBOLS)


def test_longest_prefix_specific_then_general(oracle):
    # test_ip_longest_prefix_match.rs:13-79
    db = oracle.Database(build([("192.0.2.1", {"type": "specific"}), ("192.0.2.0/24", {"type": "general"})]))
    r = db.lookup("192.0.2.1")
    assert r["kind"] == "ip" and r["prefix_len"] == 32 and r["data"] == {"type": "specific"}
    r = db.lookup("192.0.2.2")
    assert r["kind"] == "ip" and r["data"] == {"type": "general"}
    assert db.lookup("192.0.3.1")["kind"] == "notfound"


def test_longest_prefix_general_then_specific(oracle):
    # test_ip_longest_prefix_match.rs:84-150 (reverse insertion order, same answers)
    db = oracle.Database(build([("192.0.2.0/24", {"type": "general"}), ("192.0.2.1", {"type": "specific"})]))
    r = db.lookup("192.0.2.1")
    assert r["prefix_len"] == 32 and r["data"] == {"type": "specific"}
    assert db.lookup("192.0.2.200")["data"] == {"type": "general"}


def test_three_levels(oracle):
    # :155-210
    db = oracle.Database(build([("192.0.0.0/8", {"level": "8"}), ("192.0.2.1", {"level": "32"}), ("192.0.2.0/24", {"level": "24"})]))
    assert db.lookup("192.0.2.1")["data"]["level"] == "32"
    assert db.lookup("192.0.2.2")["data"]["level"] == "24"
    assert db.lookup("192.1.1.1")["data"]["level"] == "8"
    assert db.lookup("193.0.0.1")["kind"] == "notfound"


def test_ipv6_three_levels(oracle):
    # :265-320
    db = oracle.Database(build([("2001:db8::/64", {"level": "64"}), ("2001:db8::1", {"level": "128"}), ("2001:db8::/96", {"level": "96"})]))
    assert db.lookup("2001:db8::1")["data"]["level"] == "128"
    assert db.lookup("2001:db8::1")["prefix_len"] == 128
    assert db.lookup("2001:db8::2")["data"]["level"] == "96"
    assert db.lookup("2001:db8::1:0:0")["data"]["level"] == "64"
    assert db.lookup("2001:db9::1")["kind"] == "notfound"
    md = db.metadata()
    assert md["ip_version"] == 6 and md["record_size"] == 24


def test_ipv4_inside_ipv6_tree(oracle):
    # matchy-ip-trie/src/lib.rs:150-154 + tree.rs:258-277: v4 keys live under 96 zero bits
    db = oracle.Database(build([("2001:db8::1", {"v": 6}), ("10.1.2.3", {"v": 4}), ("10.9.0.0/16", {"v": 16})]))
    assert db.lookup("10.1.2.3") == {"kind": "ip", "prefix_len": 32, "data": {"v": 4}}
    r = db.lookup("10.9.77.1")
    assert r["kind"] == "ip" and r["prefix_len"] == 16 and r["data"] == {"v": 16}
    assert db.lookup("10.1.2.4")["kind"] == "notfound"
    assert db.lookup("2001:db8::1")["data"] == {"v": 6}


def test_literal_hash_roundtrip(oracle):
    # matchy-literal-hash/src/lib.rs:678-714, crates/matchy/tests/test_literal_hash.rs
    ents = [(f"pattern_{i}", {"id": i}) for i in range(100)]
    db = oracle.Database(build(ents))
    for i in range(100):
        r = db.lookup(f"pattern_{i}")
        assert r["kind"] == "pattern" and r["pattern_ids"] == [i] and r["data"] == [{"id": i}]
    assert db.lookup("pattern_100")["kind"] == "notfound"
    md = db.metadata()
    assert md["literal_entry_count"] == 100 and md["pattern_section_offset"] == 0 and md["literal_section_offset"] > 0
    assert md["database_type"] == "Paraglob-Pattern"


def test_literal_hash_many_shards(oracle):
    ents = [(f"host{i}.example{i % 97}.com", {"n": i % 5}) for i in range(12000)]
    db = oracle.Database(build(ents))
    for i in (0, 1, 4999, 11999):
        assert db.lookup(f"host{i}.example{i % 97}.com")["data"] == [{"n": i % 5}]
    assert db.lookup("host12000.example69.com")["kind"] == "notfound"


def test_paraglob_semantics(oracle):
    # paraglob_offset.rs:1890-1944 and matchy-paraglob/tests/integration_tests.rs
    db = oracle.Database(build([("*.txt", {"p": 0}), ("test_*", {"p": 1})]))
    assert db.lookup("test_file.txt")["pattern_ids"] == [0, 1]
    assert db.lookup("test_file.bin")["pattern_ids"] == [1]
    assert db.lookup("other.bin")["kind"] == "notfound"
    db = oracle.Database(build([("*", {"p": 0}), ("??", {"p": 1})]))
    assert db.lookup("ab")["pattern_ids"] == [0, 1]
    assert db.lookup("abc")["pattern_ids"] == [0]
    db = oracle.Database(build([("*test*", {"p": 0}), ("test*", {"p": 1}), ("*test", {"p": 2})]))
    assert db.lookup("test")["pattern_ids"] == [0, 1, 2]
    assert db.lookup("testing")["pattern_ids"] == [0, 1]
    assert db.lookup("mytest")["pattern_ids"] == [0, 2]
    assert db.lookup("mytesting")["pattern_ids"] == [0]
    # literal patterns inside paraglob match as substrings (Q9)
    db = oracle.Database(build([("glob:hello", {"p": 0}), ("glob:world", {"p": 1})]))
    assert db.lookup("hello world")["pattern_ids"] == [0, 1]
    # character classes
    db = oracle.Database(build([("file[0-9].txt", {"p": 0}), ("file[!0-9].txt", {"p": 1})]))
    assert db.lookup("file7.txt")["pattern_ids"] == [0]
    assert db.lookup("fileX.txt")["pattern_ids"] == [1]
    # glob whose literals are all < 3 bytes can never match (Q8)
    db = oracle.Database(build([("*.a?", {"p": 0}), ("*.evil.com", {"p": 1})]))
    assert db.lookup("x.ab")["kind"] == "notfound"
    assert db.lookup("www.evil.com")["pattern_ids"] == [1]


def test_combined_database_and_cli_style_lookup(oracle):
    # processing/mod.rs:615-702 + cli_tests.rs:946-1009
    db = oracle.Database(build([("8.8.8.8", {"who": "dns"}), ("evil.com", {"why": "bad"}), ("*.malware.com", {"why": "glob"})]))
    text = b"DNS query to evil.com from 8.8.8.8\nGET http://bad.malware.com/x\nnothing here\n"
    hits, lines, st = db.scan(text, source="test.log")
    got = [(h["type"], text[h["start"]:h["end"]].decode()) for h in hits]
    assert got == [("Domain", "evil.com"), ("IPv4", "8.8.8.8"), ("Domain", "bad.malware.com")]
    assert st.lines == 3
    recs = [json.loads(l) for l in lines]
    assert recs[0] == {"data": [{"why": "bad"}], "match_type": "pattern", "matched_text": "evil.com", "pattern_count": 1,
                       "source": "test.log", "timestamp": "0.000"}
    assert recs[1] == {"cidr": "8.8.8.8/32", "data": {"who": "dns"}, "match_type": "ip", "matched_text": "8.8.8.8",
                       "prefix_len": 32, "source": "test.log", "timestamp": "0.000"}
    assert recs[2]["matched_text"] == "bad.malware.com" and recs[2]["pattern_count"] == 1
    # key order of the emitted JSON is sorted (serde_json without preserve_order)
    assert lines[1].startswith('{"cidr":"8.8.8.8/32","data":{"who":"dns"},"match_type":"ip","matched_text":"8.8.8.8","prefix_len":32,')


def test_entry_type_detection_and_prefixes(oracle):
    db = oracle.Database(build([("literal:*.not-a-glob.com", {"k": 1}), ("glob:no-wildcards.com", {"k": 2}), ("ip:10.0.0.0/8", {"k": 3}),
                                ("[unclosed", {"k": 4})]))
    assert db.lookup("*.not-a-glob.com")["data"] == [{"k": 1}]
    assert db.lookup("x.not-a-glob.com")["kind"] == "notfound"
    assert db.lookup("www.no-wildcards.com.evil")["data"] == [{"k": 2}]   # substring semantics of paraglob literals
    assert db.lookup("10.200.1.1")["data"] == {"k": 3}
    assert db.lookup("[unclosed")["data"] == [{"k": 4}]                     # invalid glob syntax falls back to literal


def test_data_types_roundtrip(oracle):
    data = {"s": "str", "u16": 80, "u32": 70000, "u64": 5000000000, "neg": -5, "b": True, "arr": [1, "two", {"three": 3}],
            "nested": {"a": {"b": "c"}}, "dup": "str", "long": "x" * 300}
    db = oracle.Database(build([("1.2.3.4", data), ("5.6.7.8", data), ("9.9.9.9", {"s": "str"})]))
    assert db.lookup("1.2.3.4")["data"] == data
    assert db.lookup("5.6.7.8")["data"] == data
    assert db.lookup("9.9.9.9")["data"] == {"s": "str"}


def test_invalid_inputs_are_rejected():
    b = M.DatabaseBuilder()
    with pytest.raises(ValueError):
        b.add_entry("glob:[unclosed", {})
    with pytest.raises(ValueError):
        b.add_entry("ip:not-an-ip", {})
    L = M.lib()
    assert L.matchy_builder_add(None, b"x", b"{}") == -5
    assert L.matchy_builder_add(b._h, b"x", b"{not json") == -2


def test_metadata_fields(oracle):
    db = oracle.Database(build([("1.2.3.4", {"a": 1}), ("evil.com", {"a": 2}), ("*.x.org", {"a": 3})], epoch=1234567))
    md = db.metadata()
    assert md["build_epoch"] == 1234567
    assert md["database_type"] == "Paraglob-Combined-IP-Pattern"
    assert md["binary_format_major_version"] == 2 and md["languages"] == ["en"]
    assert md["ip_entry_count"] == 1 and md["literal_entry_count"] == 1 and md["glob_entry_count"] == 1
    assert md["match_mode"] == 0 and md["node_count"] == 32 and md["record_size"] == 24
    assert md["pattern_section_offset"] % 4 == 0 and md["literal_section_offset"] > md["pattern_section_offset"]


def test_validate_and_schema_entry_points(tmp_path):
    # matchy_validate (matchy.h:1357) runs without a GPU: it only parses the file
    import ctypes as C
    L = M.lib()
    good = tmp_path / "good.mxy"
    good.write_bytes(build([("10.0.0.0/8", {"a": 1}), ("*.evil.com", {"b": 2}), ("exact.example", {"c": 3})]))
    err = C.c_void_p()
    assert L.matchy_validate(str(good).encode(), 0, C.byref(err)) == 0 and not err.value
    assert L.matchy_validate(str(good).encode(), 1, None) == 0
    assert L.matchy_validate(str(good).encode(), 7, None) == -5          # unknown level -> INVALID_PARAM
    assert L.matchy_validate(None, 0, None) == -5
    bad = tmp_path / "bad.mxy"
    blob = bytearray(good.read_bytes())
    bad.write_bytes(bytes(blob[: len(blob) // 3]))                       # truncated file
    assert L.matchy_validate(str(bad).encode(), 0, C.byref(err)) == -3   # CORRUPT_DATA with a message
    assert err.value and len(C.string_at(err.value)) > 0
    L.matchy_free_string(err)
    err = C.c_void_p()
    assert L.matchy_validate(str(tmp_path / "missing.mxy").encode(), 0, C.byref(err)) == -6   # IO
    assert C.string_at(err.value) == b"Failed to validate database"
    L.matchy_free_string(err)
    # schema validation is not built: every name is unknown (MATCHY_ERROR_UNKNOWN_SCHEMA)
    b = L.matchy_builder_new()
    assert L.matchy_builder_set_schema(b, b"threatdb") == -8
    assert L.matchy_builder_set_schema(None, b"threatdb") == -5
    L.matchy_builder_free(b)
    # NULL handling of the walkers
    assert L.matchy_result_get_entry(None, None) == -5
    assert L.matchy_get_entry_data_list(None, None) == -5
    L.matchy_free_entry_data_list(None)
    L.matchy_get_stats(None, None)
    L.matchy_clear_cache(None)
    assert L.matchy_has_pattern_data(None) is False


@pytest.mark.parametrize("bits", [28, 32])
def test_wider_tree_records_read_back_identically(oracle, monkeypatch, bits):
    """28- and 32-bit MMDB records (format.rs:31-171 / tree.rs:46-130; chosen for very large trees, BASELINE configs[4]):
    the same entries written with 24-, 28- and 32-bit records answer every probe identically, IPv4 and IPv6."""
    import random
    rng = random.Random(bits)
    entries = []
    for i in range(3000):
        a = rng.getrandbits(32)
        p = rng.choice([8, 12, 16, 20, 23, 24, 24, 24, 28, 32, 32])
        a &= 0xFFFFFFFF ^ ((1 << (32 - p)) - 1)
        entries.append((f"{a >> 24}.{(a >> 16) & 255}.{(a >> 8) & 255}.{a & 255}/{p}", {"i": i}))
    for i in range(300):
        entries.append((f"2001:db8:{rng.getrandbits(16):x}::/{rng.choice([48, 56, 64])}", {"v6": i}))
    monkeypatch.delenv("MATCHY_AMD_MIN_RECORD_SIZE", raising=False)
    narrow = oracle.Database(build(entries))
    monkeypatch.setenv("MATCHY_AMD_MIN_RECORD_SIZE", str(bits))
    wide = oracle.Database(build(entries))
    assert narrow.metadata()["record_size"] == 24 and wide.metadata()["record_size"] == bits
    assert narrow.metadata()["node_count"] == wide.metadata()["node_count"]
    probes = []
    for k, _ in entries[::7]:
        probes.append(k.split("/")[0])
    for _ in range(2000):
        a = rng.getrandbits(32)
        probes.append(f"{a >> 24}.{(a >> 16) & 255}.{(a >> 8) & 255}.{a & 255}")
    probes += [f"2001:db8:{rng.getrandbits(16):x}::1" for _ in range(300)] + ["::1", "::ffff:1.2.3.4", "1.2.3.4"]
    hits = 0
    for q in probes:
        a, b = narrow.lookup(q), wide.lookup(q)
        assert a == b, q
        hits += a is not None and a.get("kind") == "ip"
    assert hits > 300


def build_ci(entries, epoch=1700000000):
    b = M.DatabaseBuilder(build_epoch=epoch, case_insensitive=True)
    for k, v in entries:
        b.add_entry(k, v)
    blob = b.build()
    b.close()
    return blob


def test_lowercase_table_matches_the_interpreter(oracle):
    """matchy_amd/data/lowercase.bin through the oracle's restatement of Rust's str::to_lowercase (per-character mapping,
    Final_Sigma rule) against Python's str.lower(), which implements the same Unicode default case conversion."""
    for s in ["HELLO", "ÉCOLE.Example.COM", "İstanbul", "ǅ", "ẞ", "K", "ΣΑΣ", "ΑΣ", "Σ", "ΑΣ.Β", "ΑΣ́", "́Σ", "aΣb", "ὈΔΥΣΣΕΎΣ", "A.Σ", "Σ.A", "",
              "mixed ÀÉÎÕÜ ΑΒΓ АБВ ԱԲԳ ᏣᎳᎩ \U00010400\U00010401 🌍"]:
        assert oracle.to_lowercase(s) == s.lower(), s
    # ... except the capitals Unicode 14.0 / 16.0 added, which the table carries by hand (tools/gen_lowercase.py) and this
    # interpreter's Unicode 13 tables map to themselves
    from tools import gen_lowercase
    added = dict(gen_lowercase.added_pairs())
    bad = [cp for cp in range(0x80, 0x30000) if not 0xD800 <= cp <= 0xDFFF and cp not in added and oracle.to_lowercase(chr(cp) + "x") != (chr(cp) + "x").lower()]
    assert not bad, [hex(c) for c in bad[:10]]
    assert len(added) == 5 + 5 + 11 + 15 + 7 + 2 + 22
    for up, lo in added.items():
        assert oracle.to_lowercase(chr(up) + "x") == chr(lo) + "x", hex(up)
        assert oracle.to_lowercase("a" + chr(lo)) == "a" + chr(lo)


def test_case_insensitive_reference_vectors(oracle):
    """The reference's own case-insensitive vectors (matchy-paraglob/tests/integration_tests.rs:99-119,
    paraglob_offset.rs:1928-1935, glob.rs:473-480, 641-648) through a database written with
    matchy_builder_set_case_insensitive and read by the oracle."""
    db = oracle.Database(build_ci([("Test*", {"p": 0}), ("glob:HELLO", {"p": 1})]))
    assert db.metadata()["match_mode"] == 1
    for q in ("Test123", "test123", "HELLO", "hello"):
        assert db.lookup(q)["kind"] == "pattern", q
    assert db.lookup("tes")["kind"] != "pattern"
    db = oracle.Database(build_ci([("glob:Hello", {"p": 0}), ("*.TXT", {"p": 1})]))
    r = db.lookup("hello test.txt")
    assert r["kind"] == "pattern" and len(r["data"]) == 2
    db = oracle.Database(build_ci([("glob:hello", {"p": 0}), ("abc[a-z]def", {"p": 1}), ("pre[!A-C]post", {"p": 2})]))
    for q, found in [("hello", True), ("HELLO", True), ("HeLLo", True), ("abcadef", True), ("abcAdef", True), ("ABCZDEF", True), ("abc1def", False),
                     ("predpost", True), ("preBpost", False), ("prebpost", False), ("PREDPOST", True)]:
        assert (db.lookup(q)["kind"] == "pattern") == found, q
    # literal keys: stored and queried through to_lowercase (matchy-literal-hash/src/lib.rs:158-167, 469-472)
    db = oracle.Database(build_ci([("Evil.Example.COM", {"k": 1}), ("ÉCOLE.example", {"k": 2}), ("ΟΔΥΣΣΕΥΣ.example", {"k": 3})]))
    for q, k in [("evil.example.com", 1), ("EVIL.EXAMPLE.COM", 1), ("école.example", 2), ("École.EXAMPLE", 2), ("οδυσσευσ.example", 3),
                 ("ΟΔΥΣΣΕΥΣ.EXAMPLE", 3)]:
        r = db.lookup(q)
        assert r["kind"] == "pattern" and r["data"] == [{"k": k}], q
    assert db.lookup("evil.example.org")["kind"] != "pattern"
    # Final_Sigma looks through the case-ignorable '.': the capital sigma before ".example" is followed by a cased letter, so the
    # key holds a medial sigma and the spelling with a final one is a different string
    assert db.lookup("οδυσσευς.example")["kind"] != "pattern"
    # globs: the AC literal is to_lowercase()d, the text only ASCII-lower-cased (matchy-ac/src/lib.rs:209,
    # paraglob_offset.rs:1198-1206): a non-ASCII capital in the text does not reach the literal
    db = oracle.Database(build_ci([("*.ÉCOLE.example", {"g": 1})]))
    assert db.lookup("www.école.EXAMPLE")["kind"] != "pattern"      # literal segment "ÉCOLE" != "école" (ASCII folding only)
    assert db.lookup("www.École.example")["kind"] != "pattern"      # AC: text keeps "É", the automaton holds "é"
    db = oracle.Database(build_ci([("*.école.Example", {"g": 1})]))
    assert db.lookup("WWW.école.EXAMPLE")["kind"] == "pattern"
    # case-sensitive databases are untouched
    db = oracle.Database(build([("Test*", {"p": 0})]))
    assert db.metadata()["match_mode"] == 0 and db.lookup("test1")["kind"] != "pattern" and db.lookup("Test1")["kind"] == "pattern"


def test_validate_rejects_structurally_corrupt_files(tmp_path):
    """matchy_validate (= what matchy_open checks before anything is uploaded): a file whose IP records, literal-hash slots, glob
    segments or Aho-Corasick nodes point outside their sections is refused with a message; the intact file passes. Offsets as
    in SURVEY.md Appendix A."""
    import ctypes as C
    import struct
    import matchy_amd as M
    from tools import synth
    blob = synth.build_db(synth.config("c1"))
    L = M.lib()

    def validate(data):
        p = tmp_path / "t.mxy"
        p.write_bytes(data)
        msg = C.c_void_p()
        rc = L.matchy_validate(str(p).encode(), 0, C.byref(msg))
        text = C.cast(msg, C.c_char_p).value.decode() if msg.value else ""
        if msg.value:
            L.matchy_free_string(msg)
        return rc, text

    assert validate(blob) == (0, "")
    u32 = lambda b, o: struct.unpack_from("<I", b, o)[0]
    # IP tree: first record of node 0 far beyond the data section
    bad = bytearray(blob); bad[0:3] = b"\xff\xff\xff"
    rc, text = validate(bytes(bad)); assert rc != 0 and "IP tree" in text
    # literal hash: the string offset of the first occupied slot moved to the last byte of the pool
    lit = blob.index(b"MMDB_LITERAL") + 16
    table_size, strings_size, shards = u32(blob, lit + 12), u32(blob, lit + 20), u32(blob, lit + 24)
    t0 = lit + 32 + (shards + 1) * 4
    slot = next(i for i in range(table_size) if u32(blob, t0 + 16 * i + 8) != 0xFFFFFFFF)
    bad = bytearray(blob); struct.pack_into("<I", bad, t0 + 16 * slot + 8, strings_size - 1)
    rc, text = validate(bytes(bad)); assert rc != 0 and "Literal hash" in text
    # paraglob: data offset of the first glob segment that carries data, wrapped to the top of the 32-bit range
    pg = blob.index(b"MMDB_PATTERN") + 16 + 8
    assert blob[pg:pg + 8] == b"PARAGLOB"
    gso, count = u32(blob, pg + 104), u32(blob, pg + 32)
    seg = None
    for pid in range(count):
        first, cnt = u32(blob, pg + gso + 8 * pid), u32(blob, pg + gso + 8 * pid + 4) & 0xFFFF
        for k in range(cnt):
            if blob[pg + first + 12 * k] in (0, 3):
                seg = pg + first + 12 * k
                break
        if seg:
            break
    bad = bytearray(blob); struct.pack_into("<I", bad, seg + 8, 0xFFFFFFF0)
    rc, text = validate(bytes(bad)); assert rc != 0 and "Glob segment" in text
    # Aho-Corasick: the root's edge table moved outside the automaton
    ac = pg + u32(blob, pg + 20)
    bad = bytearray(blob); struct.pack_into("<I", bad, ac + 12, 0xFFFFFFF0)
    rc, text = validate(bytes(bad)); assert rc != 0 and "Aho-Corasick" in text
    # wildcard count inflated
    bad = bytearray(blob); struct.pack_into("<I", bad, pg + 60, 0x40000000)
    rc, text = validate(bytes(bad)); assert rc != 0 and "Wildcard" in text
    # Aho-Corasick: a failure link that does not lead to a shallower node (here: a child of the root pointing at itself) would
    # make the device walk follow failure links for ever
    w0, eo = u32(blob, ac), u32(blob, ac + 12)
    kind = w0 & 0xFF
    child = eo if kind == 1 else u32(blob, ac + eo + 4) if kind == 2 else next(t for t in (u32(blob, ac + eo + 4 * c) for c in range(256)) if t)
    bad = bytearray(blob); struct.pack_into("<I", bad, ac + child + 8, child)
    rc, text = validate(bytes(bad)); assert rc != 0 and "failure link" in text
    # AC literal map: a literal id far beyond the table (the upload sizes its literal -> pattern index from the largest id)
    aclh = pg + u32(blob, pg + 96)
    assert blob[aclh:aclh + 4] == b"ACLH"
    slots = u32(blob, aclh + 12)
    used = next(i for i in range(slots) if u32(blob, aclh + 24 + 16 * i) != 0xFFFFFFFF)
    bad = bytearray(blob); struct.pack_into("<I", bad, aclh + 24 + 16 * used, 0xFFFFFFF0)
    rc, text = validate(bytes(bad)); assert rc != 0 and "implausible literal id" in text
    # literal mappings: a pattern id beyond the number of literals
    mstart = lit + u32(blob, lit + 16) + strings_size
    assert u32(blob, mstart) > 1
    bad = bytearray(blob); struct.pack_into("<I", bad, mstart + 4 + 8, 0x7FFFFFFF)
    rc, text = validate(bytes(bad)); assert rc != 0 and "implausible pattern id" in text


HANDMADE = ("24", "28", "32", "v6")


def _handmade():
    import json as _json
    gold = Path(__file__).parent / "golden"
    exp = _json.loads((gold / "handmade_expect.json").read_text())
    return [(name, (gold / f"handmade_{name}.mxy").read_bytes(), exp[name]) for name in HANDMADE]


def test_handmade_files_pin_the_oracle_reader(oracle):
    """tests/golden/handmade_*.mxy were assembled byte by byte from the format description by make_handmade_mxy.py — neither the
    product's builder nor the oracle wrote them. Every query's expected answer follows from the construction (24/28/32-bit
    records, IPv4 inside an IPv6 tree, a two-shard literal hash, an Aho-Corasick automaton with EMPTY / ONE / SPARSE / DENSE
    nodes, a full ACLH table that works with any slot hash, a pure-wildcard entry)."""
    for name, blob, exp in _handmade():
        odb = oracle.Database(blob)
        md = odb.metadata()
        assert md["record_size"] == exp["record_size"] and md["ip_version"] == exp["ip_version"]
        for q in exp["queries"]:
            got, want = odb.lookup(q["query"]), q["expect"]
            if want["kind"] == "pattern":
                assert got["kind"] == "pattern" and got["pattern_ids"] == want["ids"] and got["data"] == want["data"], (name, q, got)
            elif want["kind"] == "ip":
                assert got == want, (name, q, got)
            else:
                assert got == {"kind": "notfound"}, (name, q, got)


def test_handmade_files_pass_structural_validation():
    import ctypes as C
    import matchy_amd as M
    gold = Path(__file__).parent / "golden"
    for name in HANDMADE:
        msg = C.c_void_p()
        assert M.lib().matchy_validate(str(gold / f"handmade_{name}.mxy").encode(), 0, C.byref(msg)) == 0


def test_ac_literal_map_layout_for_the_reference_reader():
    """The AC literal map (ACLH) this repository writes: sized like ACLiteralHashBuilder::build (literal_hash.rs:120-160:
    max(ceil(1.25 n), 16) slots), real entries reachable from FxHasher(id) % table_size by linear probing WITHOUT crossing a
    filler (what the reference reader does when the believed hash is right), and no EMPTY slot anywhere (so the reference's probe,
    which gives up at the first empty slot, reaches every entry even if the believed hash were wrong: literal_hash.rs:263-299)."""
    import struct
    import matchy_amd as M
    for n_globs in (3, 40, 1000):
        b = M.DatabaseBuilder(build_epoch=3)
        for i in range(n_globs):
            b.add_entry(f"*.host{i}.example{i % 7}.com", {"i": i})
        blob = b.build()
        b.close()
        p = blob.index(b"MMDB_PATTERN\0\0\0\0") + 16
        pg = blob[p + 8:]
        assert pg[:8] == b"PARAGLOB"
        map_off, map_cnt = struct.unpack_from("<II", pg, 96)
        a = pg[map_off:]
        magic, ver, n, table_size, pstart, psize = struct.unpack_from("<4sIIIII", a, 0)
        assert magic == b"ACLH" and ver == 1 and n == map_cnt
        assert table_size == max((n * 5 + 3) // 4, 16)
        ents = [struct.unpack_from("<IIII", a, 24 + 16 * s) for s in range(table_size)]
        assert all(e[0] != 0xFFFFFFFF for e in ents)
        real = {e[0] for e in ents if e[2] > 0}
        assert real == set(range(n))
        assert all(e[0] >= 0x80000000 and e[1] == 0 for e in ents if e[2] == 0)

        def fx(v):
            h = (v * 0xf1357aea2e62a9c5) & 0xFFFFFFFFFFFFFFFF
            return ((h << 26) | (h >> 38)) & 0xFFFFFFFFFFFFFFFF
        worst = 0
        for lid in range(n):
            s, steps = fx(lid) % table_size, 0
            while ents[s][0] != lid:
                assert ents[s][2] > 0, "a filler sits on the probe path of a real entry"
                s = (s + 1) % table_size
                steps += 1
            worst = max(worst, steps)
        assert worst < 64


def test_lowercase_table_covers_case_pairs_added_after_unicode_13(oracle):
    """matchy_amd/data/lowercase.bin is generated from this interpreter's Unicode 13 tables plus the hand-listed additions of Unicode
    14.0 / 16.0 (tools/gen_lowercase.py); the oracle (same data file) must answer the capital spelling from a case-insensitive
    database keyed by the lower-case one. The same pairs run on the GPU in test_case_insensitive_special_casing_from_the_unicode_standard."""
    import matchy_amd as M
    pairs = [("Ⱟx.example.com", "ⱟx.example.com"), ("ꟀꟘ.example.com", "ꟁꟙ.example.com"),
             ("\U00010570\U00010595.example.com", "\U00010597\U000105BC.example.com"), ("Ᲊa.example.com", "ᲊa.example.com"),
             ("ꟋaꟜ.example.com", "ɤaƛ.example.com"), ("\U00010D50\U00010D65.example.com", "\U00010D70\U00010D85.example.com")]
    b = M.DatabaseBuilder(build_epoch=7, case_insensitive=True)
    for i, (_, low) in enumerate(pairs):
        b.add_entry(low, {"k": i})
    blob = b.build()
    b.close()
    odb = oracle.Database(blob)
    for i, (up, low) in enumerate(pairs):
        for q in (up, low):
            assert odb.lookup(q)["data"] == [{"k": i}], q
    # a key given in capitals is stored lower-cased by the builder (matchy-literal-hash/src/lib.rs:158-167)
    b = M.DatabaseBuilder(build_epoch=7, case_insensitive=True)
    b.add_entry(pairs[4][0], {"k": 1})
    assert oracle.Database(b.build()).lookup(pairs[4][1])["data"] == [{"k": 1}]
    b.close()


def _tree_kat():
    import importlib.util
    spec = importlib.util.spec_from_file_location("make_tree_kat", Path(__file__).parent / "golden" / "make_tree_kat.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.files()


def test_tree_record_vectors_of_the_reference_pin_the_oracle_reader(oracle):
    """mmdb/tree.rs:322-398 (test_read_24bit_record, test_read_28bit_record, test_calculate_data_offset) as whole files
    (tests/golden/make_tree_kat.py, assembled here: the 28- and 32-bit files are ~32 MiB because records 0x1000001 / 0x2000002
    are data pointers that far into the data section). The 28-bit node carries NON-ZERO high nibbles in its middle byte."""
    for name, (blob, node0, queries) in _tree_kat().items():
        odb = oracle.Database(blob)
        for q, want in queries:
            got = odb.lookup(q)
            if want is None:
                assert got == {"kind": "notfound"}, (name, q, got)
            else:
                assert got == {"kind": "ip", "prefix_len": want[0], "data": want[1]}, (name, q, got)
        odb.close()


def _glob_kat_dbs():
    """[(case, blob)]: one database per vector of tests/golden/glob_kat.json holding the pattern as its only key"""
    import json as _json
    import matchy_amd as M
    cases = _json.loads((Path(__file__).parent / "golden" / "glob_kat.json").read_text())["cases"]
    out = []
    for c in cases:
        b = M.DatabaseBuilder(build_epoch=1, case_insensitive=c["case_insensitive"])
        b.add_entry(c["pattern"], {"p": 1})
        out.append((c, b.build()))
        b.close()
    return out


def test_glob_vectors_of_the_reference_pin_the_oracle(oracle):
    """glob.rs:464-705 (`GlobPattern::matches`: stars, question marks, classes, negation, ranges, case folding, multi-byte characters) through a
    database whose only key is the pattern — the cases where the database layer answers like GlobPattern (tests/golden/make_glob_kat.py says
    which were left out and why)."""
    n = 0
    for c, blob in _glob_kat_dbs():
        odb = oracle.Database(blob)
        for t in c["match"]:
            assert odb.lookup(t)["kind"] == "pattern", (c["ref"], c["pattern"], t)
        for t in c["nomatch"]:
            assert odb.lookup(t)["kind"] == "notfound", (c["ref"], c["pattern"], t)
        n += len(c["match"]) + len(c["nomatch"])
        odb.close()
    assert n >= 60


# crates/matchy/tests/test_ip_exact_match.rs, transcribed as data: (reference lines, keys, addresses that must be found, addresses that must not)
IP_EXACT_MATCH_KAT = [
    ("test_ip_exact_match.rs:13-86", [("0.0.0.1", {}), ("0.0.0.3", {}), ("0.0.0.5", {})],
     ["0.0.0.1", "0.0.0.3", "0.0.0.5"], ["0.0.0.0", "0.0.0.2", "0.0.0.4", "0.0.0.6"]),
    ("test_ip_exact_match.rs:89-138", [(f"10.0.0.{i}", {}) for i in range(10)],
     [f"10.0.0.{i}" for i in range(10)], [f"10.0.0.{i}" for i in range(10, 20)] + ["10.0.1.0"]),
    ("test_ip_exact_match.rs:141-196", [("192.168.1.1", {}), ("192.168.1.100", {}), ("192.168.1.200", {})],
     ["192.168.1.1", "192.168.1.100", "192.168.1.200"], ["192.168.1.2", "192.168.1.50", "192.168.1.150", "192.168.1.250"]),
    ("test_ip_exact_match.rs:199-270", [("10.0.0.0/30", {"type": "cidr"}), ("10.0.0.5", {"type": "individual"}), ("10.0.0.10", {"type": "individual"})],
     ["10.0.0.0", "10.0.0.1", "10.0.0.2", "10.0.0.3", "10.0.0.5", "10.0.0.10"], ["10.0.0.4", "10.0.0.7"]),
]


def test_exact_ip_matching_vectors_of_the_reference(oracle):
    """test_ip_exact_match.rs: individually inserted addresses match exactly — no neighbour, no implied range — and a /30 beside them covers
    its four addresses only. Oracle over builder-made databases."""
    for ref, entries, found, missing in IP_EXACT_MATCH_KAT:
        db = oracle.Database(build(entries))
        for q in found:
            assert db.lookup(q)["kind"] == "ip", (ref, q)
        for q in missing:
            assert db.lookup(q)["kind"] == "notfound", (ref, q)
        db.close()


# crates/matchy/tests/test_literal_hash.rs:52-300 as data: (reference lines, keys, [(query, kind the reference asserts, number of pattern ids)]).
# add_literal / add_glob of the reference are the `literal:` / `glob:` prefixes of add_entry (mmdb_builder.rs:399-406); a literal holding glob
# characters stays a literal (:220-265). The last row is test_builder_stats (:267-300) read back from the metadata the builder writes.
LITERAL_HASH_KAT = [
    ("test_literal_hash.rs:53", [("literal:evil.com", {"source": "literal"}), ("glob:*.com", {"source": "glob"})],
     [("evil.com", "pattern", 2), ("other.com", "pattern", 1), ("evil.org", "notfound", 0)]),
    ("test_literal_hash.rs:118", [("glob:*.phishing.com", {"type": "glob"}), ("glob:bad-*", {"type": "glob"})],
     [("test.phishing.com", "pattern", 1), ("bad-actor", "pattern", 1), ("good-actor", "notfound", 0)]),
    ("test_literal_hash.rs:161", [("1.2.3.4", {"type": "ip"}), ("literal:evil.com", {"type": "literal"}), ("glob:*.bad.com", {"type": "glob"})],
     [("1.2.3.4", "ip", 0), ("evil.com", "pattern", 1), ("test.bad.com", "pattern", 1), ("1.2.3.5", "notfound", 0)]),
    ("test_literal_hash.rs:221", [("literal:file[1].txt", {"note": "has brackets"}), ("literal:what?.com", {"note": "has brackets"}),
                                  ("literal:price*list", {"note": "has brackets"})],
     [("file[1].txt", "pattern", 1), ("what?.com", "pattern", 1), ("price*list", "pattern", 1), ("file2.txt", "notfound", 0),
      ("file1.txt", "notfound", 0), ("whatX.com", "notfound", 0), ("priceXXlist", "notfound", 0)]),
    ("test_literal_hash.rs:268", [("1.2.3.4", {}), ("literal:evil.com", {}), ("literal:bad.org", {}), ("glob:*.phishing.com", {})], []),
]


def test_literal_hash_vectors_of_the_reference(oracle):
    for ref, entries, checks in LITERAL_HASH_KAT:
        db = oracle.Database(build(entries))
        for q, kind, n in checks:
            r = db.lookup(q)
            assert r["kind"] == kind and len(r.get("pattern_ids", [])) == n, (ref, q, r)
        if ref.endswith(":53"):
            assert sorted(d["source"] for d in db.lookup("evil.com")["data"]) == ["glob", "literal"]
        if ref.endswith(":268"):
            md = db.metadata()
            assert (md["ip_entry_count"], md["literal_entry_count"], md["glob_entry_count"]) == (1, 2, 1)
        db.close()


# crates/matchy-paraglob/tests/integration_tests.rs:11-260 (Paraglob::find_all on patterns given directly to the paraglob builder), as data.
# Through a database the same patterns are written with the `glob:` prefix where they carry no wildcard — that is what routes a key into the
# paraglob section as a LITERAL pattern with its substring semantics (mmdb_builder.rs:399-406, Q9); pattern ids are positional (all keys are
# glob entries). Expectation per text: ("n", k) exactly k ids, ("ge", k) at least k, ("has", [ids]) those ids among them, ("none",) nothing.
# Not transcribed: test_duplicate_pattern_deduplication (:43-62: duplicate strings inside ONE paraglob builder — the database builder keys its
# entries by string) and the `v2` tests (:263-700: the paraglob's own data section, unused in combined databases).
PARAGLOB_KAT = [
    ("integration_tests.rs:11", False, ["*.txt", "test*", "*file*"],
     [("document.txt", ("ge", 1)), ("test_case", ("ge", 1)), ("myfile.dat", ("ge", 1)), ("nomatch", ("none",))]),
    ("integration_tests.rs:27", False, ["hello", "world", "test"],
     [("hello", ("n", 1)), ("world", ("n", 1)), ("hello world", ("n", 2)), ("nomatch", ("none",))]),
    ("integration_tests.rs:64", False, ["*.txt", "*file*", "test*"], [("testfile.txt", ("has", [0, 1, 2])), ("testfile.txt", ("n", 3))]),
    ("integration_tests.rs:78", False, ["Test*", "HELLO"], [("Test123", ("ge", 1)), ("test123", ("none",)), ("HELLO", ("ge", 1)), ("hello", ("none",))]),
    ("integration_tests.rs:100", True, ["Test*", "HELLO"], [("Test123", ("ge", 1)), ("test123", ("ge", 1)), ("HELLO", ("ge", 1)), ("hello", ("ge", 1))]),
    ("integration_tests.rs:122", False, ["test"], [("", ("none",)), ("test", ("n", 1))]),
    ("integration_tests.rs:134", False, ["exact_match", "another_literal", "third"],
     [("exact_match", ("n", 1)), ("prefix_exact_match_suffix", ("n", 1)), ("nomatch", ("none",))]),
    ("integration_tests.rs:152", False, ["*test*", "test*", "*test"], [("test", ("n", 3)), ("testing", ("n", 2)), ("mytest", ("n", 2)), ("mytesting", ("n", 1))]),
    ("integration_tests.rs:169", False, ["*.rs", "*.toml", "Cargo.*", "src/*", "*.md"],
     [("main.rs", ("ge", 1)), ("Cargo.toml", ("ge", 2)), ("src/lib.rs", ("ge", 2)), ("README.md", ("ge", 1)), ("test.py", ("none",))]),
    ("integration_tests.rs:190", False, [f"pattern_{i}_*" for i in range(1000)],
     [("pattern_500_test", ("has", [500])), ("pattern_999_data", ("has", [999])), ("nomatch", ("none",))]),
    ("integration_tests.rs:222", False, ["hello", "*.txt", "test_*"],
     [("hello.txt", ("n", 2)), ("hello.txt", ("has", [0, 1])), ("test_file.txt", ("n", 2)), ("test_file.txt", ("has", [1, 2]))]),
    ("integration_tests.rs:242", False, ["*", "?", "**"], [("test", ("ge", 2)), ("a", ("ge", 3)), ("", ("ge", 2))]),
]


def paraglob_kat_db(patterns, ci):
    b = M.DatabaseBuilder(build_epoch=9, case_insensitive=ci)
    for i, p in enumerate(patterns):
        wild = any(ch in p for ch in "*?[")
        b.add_entry(p if wild else "glob:" + p, {"i": i})
    blob = b.build()
    b.close()
    return blob


def paraglob_kat_check(ids, expect):
    if expect[0] == "n":
        return len(ids) == expect[1]
    if expect[0] == "ge":
        return len(ids) >= expect[1]
    if expect[0] == "has":
        return all(i in ids for i in expect[1])
    return not ids


def test_paraglob_integration_vectors_of_the_reference(oracle):
    for ref, ci, patterns, checks in PARAGLOB_KAT:
        db = oracle.Database(paraglob_kat_db(patterns, ci))
        for text, expect in checks:
            r = db.lookup(text)
            ids = r.get("pattern_ids", []) if r["kind"] == "pattern" else []
            assert paraglob_kat_check(ids, expect), (ref, text, expect, ids)
        db.close()
