"""CPU: the oracle restatement against the known-answer vectors the reference's own tests hold
(tests/golden/extractor_kat.json; transcribed by tests/golden/make_extractor_kat.py)."""
import json
from pathlib import Path

import pytest

GOLD = Path(__file__).parent / "golden"
CASES = json.loads((GOLD / "extractor_kat.json").read_text())["cases"]


def _input(c):
    return bytes.fromhex(c["input_hex"]) if "input_hex" in c else c["input"].encode("utf-8")


@pytest.mark.parametrize("c", CASES, ids=[f"{i}:{c['ref']}" for i, c in enumerate(CASES)])
def test_extractor_kat(oracle, c):
    data = _input(c)
    got = oracle.extract(data, flags=c["flags"], min_labels=c["min_labels"])
    by_type = {}
    for t, s, e, v in got:
        by_type.setdefault(t, []).append(v)
    for t, want in c["expect"].items():
        assert by_type.get(t, []) == want, (t, got)
    if "total" in c:
        assert len(got) == c["total"], got
    for t, sub in c.get("forbid_substring", {}).items():
        assert not any(sub in v for v in by_type.get(t, []))
    for t, suf in c.get("forbid_suffix", {}).items():
        assert not any(v.endswith(suf) for v in by_type.get(t, []))


def test_primitives_kat(oracle):
    # XXH64 spec vectors, FIPS 180-4 / Keccak team vectors
    assert oracle.xxh64(b"") == 0xEF46DB3751D8E999
    assert oracle.xxh64(b"abc") == 0x44BC2CF5AD770999
    assert oracle.xxh64(b"evil.com") == 0x4A37AA533DBB4AE5
    import xxhash
    for n in (1, 3, 4, 7, 8, 15, 31, 32, 33, 63, 64, 100, 253):
        b = bytes((i * 7 + 3) & 0xFF for i in range(n))
        assert oracle.xxh64(b) == xxhash.xxh64(b, seed=0).intdigest()
    import hashlib
    for msg in (b"", b"abc", b"a" * 55, b"a" * 56, b"a" * 64, b"a" * 119, b"a" * 1000):
        assert oracle.sha256(msg) == hashlib.sha256(msg).digest()
    assert oracle.keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
    assert oracle.keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
    assert oracle.keccak256(b"a" * 136).hex() != oracle.keccak256(b"a" * 135).hex()


def test_psl_container(oracle):
    L = oracle.lib()
    assert L.orc_psl_count() == 10496
    for s, want in ((b"com", 1), (b"co.uk", 1), (b"community", 1), (b"peer", 0), (b"COM", 0), (b"*.ck", 1), (b"4", 0), (b"html", 0)):
        assert L.orc_psl_contains(s, len(s)) == want, s


def test_ipv6_display_and_parse(oracle):
    import ctypes as C
    L = oracle.lib()

    def parse(s):
        out = C.create_string_buffer(16)
        ok = L.orc_parse_ipv6(s.encode(), len(s), out)
        return oracle.format_ip(out.raw, True) if ok else None

    assert parse("2001:0db8::1") == "2001:db8::1"
    assert parse("2001:db8:0:0:1:0:0:1") == "2001:db8::1:0:0:1"
    assert parse("1:2:3:4:5:6:7::") == "1:2:3:4:5:6:7:0"
    assert parse("1:2:3:4::5:6:7:8") is None      # "::" must stand for at least one group
    assert parse("1:2:3:4:5:6:7:8") == "1:2:3:4:5:6:7:8"
    assert parse("::ffff:102:304") == "::ffff:1.2.3.4"
    assert parse("12345::1") is None
    assert parse("1::2::3") is None
    assert parse("1:::2") is None
    assert parse("::") == "::"


GOLDEN_CONFIGS = [("c1", 10000), ("c2", 1000), ("c3", 1000), ("c3b", 1000), ("c4", 1000), ("c5/100", 1000)]


def _golden(cfgname):
    import json
    from pathlib import Path
    p = Path(__file__).parent / "golden" / f"config_{cfgname.replace('/', '_')}.ndjson"
    rows = p.read_text().splitlines()
    return json.loads(rows[0]), rows[1:]


@pytest.mark.parametrize("cfgname,lines", GOLDEN_CONFIGS)
def test_oracle_reproduces_config_golden(oracle, cfgname, lines):
    """tests/golden/config_*.ndjson (tests/golden/make_config_golden.py): the BASELINE configs' match sets — database built by
    the product's builder, log from the counter-based generator, scan by the oracle — stay what they were when committed."""
    from tools import synth
    head, want = _golden(cfgname)
    cfg = synth.config(cfgname)
    hits, ndjson, st = oracle.Database(synth.build_db(cfg)).scan(synth.make_log(cfg, 0, lines), source="access.log")
    assert (st.lines, st.candidates, len(ndjson)) == (head["lines"], head["candidates"], head["matches"])
    assert ndjson == want


def test_monero_constructed_accepts_and_rejects(oracle):
    """E9: checksum-VALID Monero tokens constructed from the rule itself (tests/golden/make_xmr_kat.py: whole-string Base58 +
    Keccak-256, crates/matchy-extractor/src/lib.rs:1367-1409, 1895-1920; the reference's own test asserts nothing on accept)."""
    k = json.loads((GOLD / "xmr_kat.json").read_text())
    assert len(k["accept"]) >= 8 and {len(a) for a in k["accept"]} >= {95, 106} and {a[0] for a in k["accept"]} == {"4", "8"}
    for a in k["accept"]:
        for pre, post in ((b"pay ", b" now"), (b"", b""), (b"addr=", b"\n"), (b"[", b"]")):
            buf = pre + a.encode() + post
            got = [m for m in oracle.extract(buf) if m[0] == "Monero"]
            assert got == [("Monero", len(pre), len(pre) + len(a), a)], buf
        # glued to another token character the token is longer than the address: not a candidate, or a failing checksum
        assert not [m for m in oracle.extract(b"x" + a.encode()) if m[0] == "Monero"]
    for r in k["reject"]:
        buf = b"pay " + r["text"].encode() + b" now"
        assert not [m for m in oracle.extract(buf) if m[0] == "Monero"], r
