#!/usr/bin/env python3
"""Assembles tiny .mxy databases BYTE BY BYTE from the on-disk format description (SURVEY.md Appendix A and the struct
definitions it cites: crates/matchy-paraglob/src/offset_format.rs:73-476, matchy-ac/src/lib.rs:74-124,
matchy-paraglob/src/literal_hash.rs:48-77, matchy-literal-hash/src/lib.rs:80-111 + 262-351, mmdb_builder.rs:573-757,
MaxMind DB data encoding matchy-data-format/src/lib.rs:374-428) — without the product's builder (matchy_amd/csrc/db_builder.cpp)
and without the oracle. The files pin the READERS (product and oracle) with bytes neither of them wrote; the expected answer of
every query below follows from how the file was put together (stated next to each query), cross-checked here with Python's
own glob matcher.

Writes tests/golden/handmade_{24,28,32,v6}.mxy and handmade_expect.json. Deliberately simple where the format allows a
choice:
  * IP tree: plain binary trie, only non-nested prefixes (no backfill quirks), leaves at depth == prefix length;
  * literal hash (LHSH): TWO shards (the reference never writes fewer than 16; the reader takes the count from the header),
    XXH64 slots computed with the `xxhash` module;
  * PARAGLOB: Aho-Corasick automaton built by the textbook algorithm (goto trie, BFS failure links, outputs merged along
    failure links) with an EMPTY, ONE, SPARSE and a DENSE (root, 9 first bytes) node kind; the AC literal -> pattern table
    (ACLH) is completely full, so a probe finds every entry whatever the slot hash function is (rustc-hash's FxHasher is
    unverified here, SURVEY §8c: a reader must not depend on it for these files).
"""
import fnmatch
import json
import struct
from pathlib import Path

import xxhash

HERE = Path(__file__).resolve().parent


# ------------------------------------------------------------------------------------------------ MaxMind data encoding
def enc_ctrl(t, size):
    if t <= 7:
        first, ext = t << 5, b""
    else:
        first, ext = 0, bytes([t - 7])
    if size < 29:
        return bytes([first | size]) + ext
    if size < 29 + 256:
        return bytes([first | 29]) + ext + bytes([size - 29])
    if size < 285 + 65536:
        return bytes([first | 30]) + ext + struct.pack(">H", size - 285)
    return bytes([first | 31]) + ext + struct.pack(">I", size - 65821)[1:]


def enc_uint(t, v):
    b = v.to_bytes((v.bit_length() + 7) // 8, "big")
    return enc_ctrl(t, len(b)) + b


class U16(int): pass
class U32(int): pass
class U64(int): pass


def enc(v):
    if isinstance(v, str):
        b = v.encode()
        return enc_ctrl(2, len(b)) + b
    if isinstance(v, U16): return enc_uint(5, int(v))
    if isinstance(v, U32): return enc_uint(6, int(v))
    if isinstance(v, U64): return enc_uint(9, int(v))
    if isinstance(v, int): return enc_uint(6, v)
    if isinstance(v, dict):
        out = enc_ctrl(7, len(v))
        for k in sorted(v):
            out += enc(k) + enc(v[k])
        return out
    if isinstance(v, list):
        return enc_ctrl(11, len(v)) + b"".join(enc(x) for x in v)
    raise TypeError(v)


# ------------------------------------------------------------------------------------------------ IP tree
def build_tree(prefixes, ip_version, record_size):
    """prefixes: [(bits as '0101...', data_offset)] non-nested. Returns (tree bytes, node_count)."""
    nodes = [[None, None]]
    for bits, off in prefixes:
        if ip_version == 6 and len(bits) <= 32 and bits.startswith("v4:"):
            raise AssertionError
        n = 0
        for i, c in enumerate(bits):
            b = int(c)
            last = i == len(bits) - 1
            if last:
                assert nodes[n][b] is None
                nodes[n][b] = ("data", off)
            else:
                if nodes[n][b] is None:
                    nodes.append([None, None])
                    nodes[n][b] = ("node", len(nodes) - 1)
                assert nodes[n][b][0] == "node"
                n = nodes[n][b][1]
    nc = len(nodes)

    def rec(x):
        if x is None:
            return nc                      # "not found"
        return x[1] if x[0] == "node" else nc + 16 + x[1]
    out = bytearray()
    for l, r in nodes:
        a, b = rec(l), rec(r)
        if record_size == 24:
            out += a.to_bytes(3, "big") + b.to_bytes(3, "big")
        elif record_size == 28:
            out += (a & 0xFFFFFF).to_bytes(3, "big") + bytes([((a >> 24) << 4) | (b >> 24)]) + (b & 0xFFFFFF).to_bytes(3, "big")
        else:
            out += a.to_bytes(4, "big") + b.to_bytes(4, "big")
    return bytes(out), nc


def v4_bits(addr, plen):
    a = 0
    for p in addr.split("."):
        a = (a << 8) | int(p)
    return format(a, "032b")[:plen]


# ------------------------------------------------------------------------------------------------ literal hash (LHSH)
def build_lhsh(literals):
    """literals: [(text, pattern_id, data_offset)] -> section bytes. Two shards, capacity 4 each (power of two)."""
    num_shards, cap = 2, 4
    shards = [[None] * cap for _ in range(num_shards)]
    pool = bytearray()
    for text, pid, _ in literals:
        raw = text.encode()
        h = xxhash.xxh64_intdigest(raw, 0)
        so = len(pool)
        pool += struct.pack("<H", len(raw)) + raw + b"\0"
        sh = shards[h % num_shards]
        s = h & (cap - 1)
        while sh[s] is not None:
            s = (s + 1) & (cap - 1)
        sh[s] = (h, so, pid)
    table = b""
    for sh in shards:
        for e in sh:
            table += struct.pack("<QII", 0, 0xFFFFFFFF, 0) if e is None else struct.pack("<QII", *e)
    shard_offsets = struct.pack("<III", 0, cap, 2 * cap)
    strings_offset = 32 + len(shard_offsets) + len(table)
    hdr = b"LHSH" + struct.pack("<IIIIIII", 1, len(literals), num_shards * cap, strings_offset, len(pool), num_shards, 1)
    maps = struct.pack("<I", len(literals)) + b"".join(struct.pack("<II", pid, off) for _, pid, off in literals)
    return hdr + shard_offsets + table + bytes(pool) + maps


# ------------------------------------------------------------------------------------------------ PARAGLOB
def parse_glob(p):
    """-> [(type, flags, payload)]: 0 literal bytes, 1 star, 2 question, 3 class [(kind, c1, c2)]."""
    segs, lit, i = [], bytearray(), 0
    def flush():
        if lit:
            segs.append((0, 0, bytes(lit)))
            lit.clear()
    while i < len(p):
        c = p[i]
        if c == "*":
            flush(); segs.append((1, 0, b"")); i += 1
        elif c == "?":
            flush(); segs.append((2, 0, b"")); i += 1
        elif c == "[":
            flush()
            j = p.index("]", i + 1)
            body, neg = p[i + 1:j], 0
            if body[0] in "!^":
                neg, body = 1, body[1:]
            items, k = [], 0
            while k < len(body):
                if k + 2 < len(body) and body[k + 1] == "-":
                    items.append((1, ord(body[k]), ord(body[k + 2]))); k += 3
                else:
                    items.append((0, ord(body[k]), 0)); k += 1
            segs.append((3, neg, items)); i = j + 1
        else:
            lit += c.encode(); i += 1
    flush()
    return segs


def build_ac(literals):
    """literals: [bytes] (id = index). Returns (AC buffer, node count). Node = ACNodeHot 20 B: kind u8, one_char u8,
    edge_count u8, pattern_count u8, one_target u32, failure_offset u32, edges_offset u32, patterns_offset u32."""
    goto, out, fail = [{}], [[]], [0]
    for lid, lit in enumerate(literals):
        s = 0
        for ch in lit:
            if ch not in goto[s]:
                goto.append({}); out.append([]); fail.append(0)
                goto[s][ch] = len(goto) - 1
            s = goto[s][ch]
        out[s].append(lid)
    order, q = [], list(goto[0].values())
    while q:
        s = q.pop(0)
        order.append(s)
        for ch, t in goto[s].items():
            f = fail[s]
            while f and ch not in goto[f]:
                f = fail[f]
            fail[t] = goto[f][ch] if ch in goto[f] and goto[f][ch] != t else 0
            out[t] = out[t] + [x for x in out[fail[t]] if x not in out[t]]
            q.append(t)
    n = len(goto)
    node_off = [20 * i for i in range(n)]
    buf = bytearray(20 * n)
    extra = bytearray()
    base = 20 * n

    def alloc(data, align=4):
        while (base + len(extra)) % align:
            extra.append(0)
        o = base + len(extra)
        extra.extend(data)
        return o
    kinds = set()
    for s in range(n):
        edges = sorted(goto[s].items())
        kind, one_char, edges_offset = 0, 0, 0
        if len(edges) == 1:
            kind, one_char, edges_offset = 1, edges[0][0], node_off[edges[0][1]]
        elif 2 <= len(edges) <= 8:
            kind = 2
            edges_offset = alloc(b"".join(struct.pack("<B3xI", ch, node_off[t]) for ch, t in edges))
        elif len(edges) >= 9:
            kind = 3
            tab = [0] * 256
            for ch, t in edges:
                tab[ch] = node_off[t]
            edges_offset = alloc(struct.pack("<256I", *tab), 64)
        kinds.add(kind)
        po = alloc(b"".join(struct.pack("<I", x) for x in out[s])) if out[s] else 0
        struct.pack_into("<BBBBIIII", buf, node_off[s], kind, one_char, min(len(edges), 255), min(len(out[s]), 255),
                         edges_offset if kind == 1 else 0, node_off[fail[s]], edges_offset, po)
    assert kinds == {0, 1, 2, 3}, kinds
    return bytes(buf) + bytes(extra), n


def build_paraglob(patterns):
    """patterns: [(text, type 0 literal/1 glob, ac literal or None)] -> (PARAGLOB buffer)."""
    ac_lits, lit_to_pats = [], {}
    for pid, (text, ptype, lit) in enumerate(patterns):
        if lit is None:
            continue
        b = lit.encode()
        if b not in ac_lits:
            ac_lits.append(b)
        lit_to_pats.setdefault(ac_lits.index(b), []).append(pid)
    ac, n_nodes = build_ac(ac_lits)
    buf = bytearray(128)                      # header (112) padded to 128
    ac_off = len(buf)
    buf += ac
    while len(buf) % 8:
        buf.append(0)
    patterns_off = len(buf)
    buf += bytes(16 * len(patterns))
    strings_off = len(buf)
    for pid, (text, ptype, _) in enumerate(patterns):
        so = len(buf)
        raw = text.encode()
        buf += raw + b"\0"
        struct.pack_into("<IB3xII", buf, patterns_off + 16 * pid, pid, ptype, so, len(raw))
    strings_size = len(buf) - strings_off
    while len(buf) % 8:
        buf.append(0)
    wild = [(pid, struct.unpack_from("<I", buf, patterns_off + 16 * pid + 8)[0]) for pid, (t, pt, lit) in enumerate(patterns) if lit is None]
    for pid, so in wild:
        buf += struct.pack("<II", pid, so)
    # ACLH: completely full table (every probe sequence visits every entry), lists behind it
    aclh_off = len(buf)
    n = len(lit_to_pats)
    lists = bytearray()
    entries = b""
    for lid in sorted(lit_to_pats):
        entries += struct.pack("<IIII", lid, len(lists), len(lit_to_pats[lid]), 0)
        lists += b"".join(struct.pack("<I", p) for p in lit_to_pats[lid])
    buf += b"ACLH" + struct.pack("<IIIII", 1, n, n, 24 + len(entries), len(lists)) + entries + lists
    while len(buf) % 8:
        buf.append(0)
    # glob segments: index, headers, then literal bytes / class items
    gso = len(buf)
    buf += bytes(8 * len(patterns))
    parsed = [parse_glob(t) for t, _, _ in patterns]
    hdr_off = []
    for pid, segs in enumerate(parsed):
        struct.pack_into("<IHH", buf, gso + 8 * pid, len(buf), len(segs), 0)
        hdr_off.append(len(buf))
        buf += bytes(12 * len(segs))
    for pid, segs in enumerate(parsed):
        for k, (st, fl, payload) in enumerate(segs):
            data = b""
            if st == 0:
                data = payload
            elif st == 3:
                while len(buf) % 4:
                    buf.append(0)
                data = b"".join(struct.pack("<B3xII", kind, c1, c2) for kind, c1, c2 in payload)
            off = len(buf) if data else 0
            buf += data
            struct.pack_into("<BBHII", buf, hdr_off[pid] + 12 * k, st, fl, 0, len(data), off)
    seg_size = len(buf) - gso
    hdr = b"PARAGLOB" + struct.pack("<26I", 5, 0, n_nodes, ac_off, len(ac), 0, len(patterns), patterns_off, strings_off, strings_size,
                                      0, 0, 0, len(wild), len(buf), 1, 0, 0, 0, 0, 0, 0, aclh_off, n, gso, seg_size)
    assert len(hdr) == 112
    buf[:112] = hdr
    return bytes(buf)


# ------------------------------------------------------------------------------------------------ whole file
def assemble(record_size, ip_version):
    # data section: one map per entry
    data = bytearray()
    offs = {}
    def put(name, value):
        offs[name] = len(data)
        data.extend(enc(value))
    for name, value in (("net10", {"n": U16(1), "who": "ten"}), ("net192", {"n": U16(2)}), ("host", {"n": U16(3)}), ("v6", {"n": U16(6)}),
                        ("lit_a", {"l": "a"}), ("lit_b", {"l": "b"}), ("lit_c", {"l": "c"}),
                        ("g0", {"g": U16(0)}), ("g1", {"g": U16(1)}), ("g2", {"g": U16(2)}), ("g3", {"g": U16(3)}), ("g4", {"g": U16(4)}),
                        ("g5", {"g": U16(5)}), ("g6", {"g": U16(6)}), ("g7", {"g": U16(7)}), ("g8", {"g": U16(8)}), ("g9", {"g": U16(9)}),
                        ("g10", {"g": U16(10)}), ("g11", {"g": U16(11)})):
        put(name, value)
    v4 = [("10.0.0.0", 8, "net10"), ("192.168.1.0", 24, "net192"), ("203.0.113.77", 32, "host")]
    prefixes = []
    for addr, plen, name in v4:
        bits = v4_bits(addr, plen)
        if ip_version == 6:
            bits = "0" * 96 + bits       # ::a.b.c.d: the reader takes 96 left steps for IPv4 queries (tree.rs:258-277)
        prefixes.append((bits, offs[name]))
    if ip_version == 6:
        prefixes.append((format(0x20010DB8, "032b"), offs["v6"]))      # 2001:db8::/32
    tree, node_count = build_tree(prefixes, ip_version, record_size)
    literals = [("evil.example.com", 0, offs["lit_a"]), ("bad-host.example.net", 1, offs["lit_b"]),
                ("d41d8cd98f00b204e9800998ecf8427e", 2, offs["lit_c"])]
    # globs: (pattern, type, AC literal = a substring every matching text contains; None = pure wildcard)
    patterns = [("*.alpha-evil.com", 1, "alpha-evil.com"), ("beta-*.net", 1, "beta-"), ("cdn[0-9].gamma.org", 1, "cdn"),
                ("delta?.io", 1, "delta"), ("*epsilon*", 1, "epsilon"), ("*.foxtrot.co.uk", 1, "foxtrot.co.uk"), ("golf.*.com", 1, "golf."),
                ("hotel-[!a-c]*.org", 1, "hotel-"), ("india.example.com", 0, "india.example.com"), ("*.alpha-evil.co", 1, ".alpha-evil.co"),
                ("be*.net", 1, None),   # this one sits in the pure-wildcard array: the reader verifies those against every text
                ("dexter-*.biz", 1, "dexter-")]   # shares "de" with "delta": that node has two edges (SPARSE)
    # ten different first bytes: the root is a DENSE node; ".alpha-evil.co" fails over into the "alpha-evil.co(m)" branch
    pg = build_paraglob(patterns)
    glob_data = [offs["g%d" % i] for i in range(len(patterns))]
    body = bytearray(tree) + bytes(16) + bytes(data)
    pad = (4 - ((len(body) + 16) % 4)) % 4
    body += bytes(pad)
    pat_off = len(body) + 16
    body += b"MMDB_PATTERN\0\0\0\0"
    sect = struct.pack("<II", 8 + len(pg) + 4 + 4 * len(glob_data), len(pg)) + pg + struct.pack("<I", len(glob_data)) + b"".join(struct.pack("<I", o) for o in glob_data)
    body += sect
    lit_off = len(body) + 16
    body += b"MMDB_LITERAL\0\0\0\0" + build_lhsh(literals)
    meta = {"binary_format_major_version": U16(2), "binary_format_minor_version": U16(0), "build_epoch": U64(1700000000),
            "database_type": "Handmade-Fixture", "description": {"en": "assembled byte by byte by tests/golden/make_handmade_mxy.py"},
            "languages": ["en"], "ip_version": U16(ip_version), "node_count": U32(node_count), "record_size": U16(record_size),
            "ip_entry_count": U32(len(prefixes)), "literal_entry_count": U32(len(literals)), "glob_entry_count": U32(len(patterns)),
            "match_mode": U16(0), "pattern_section_offset": U32(pat_off), "literal_section_offset": U32(lit_off)}
    body += b"\xAB\xCD\xEFMaxMind.com" + enc(meta)
    return bytes(body), patterns, literals


def expectations(ip_version, patterns):
    """(query, expected) — expected: {"kind": "ip", "prefix_len", "data"} | {"kind": "pattern", "ids", "data"} | {"kind": "notfound"}."""
    ex = []
    add = lambda q, e, why: ex.append({"query": q, "expect": e, "why": why})
    vp = 0    # an IPv4 query of an IPv6 tree reports the IPv4 prefix: depth - 96 + 1 (tree.rs:76-84)
    add("10.1.2.3", {"kind": "ip", "prefix_len": 8 + vp, "data": {"n": 1, "who": "ten"}}, "inside 10.0.0.0/8, leaf at depth 8")
    add("192.168.1.200", {"kind": "ip", "prefix_len": 24 + vp, "data": {"n": 2}}, "inside 192.168.1.0/24")
    add("203.0.113.77", {"kind": "ip", "prefix_len": 32 + vp, "data": {"n": 3}}, "the /32 itself")
    add("203.0.113.78", {"kind": "notfound"}, "sibling of the /32")
    add("192.168.2.1", {"kind": "notfound"}, "outside the /24")
    add("11.0.0.1", {"kind": "notfound"}, "outside the /8")
    if ip_version == 6:
        add("2001:db8::1", {"kind": "ip", "prefix_len": 32, "data": {"n": 6}}, "inside 2001:db8::/32")
        add("2001:db9::1", {"kind": "notfound"}, "outside the IPv6 prefix")
    add("evil.example.com", {"kind": "pattern", "ids": [0], "data": [{"l": "a"}]}, "literal 0")
    add("bad-host.example.net", {"kind": "pattern", "ids": [1], "data": [{"l": "b"}]}, "literal 1")
    add("d41d8cd98f00b204e9800998ecf8427e", {"kind": "pattern", "ids": [2], "data": [{"l": "c"}]}, "literal 2 (other shard or probe)")
    add("Evil.example.com", {"kind": "notfound"}, "case-sensitive database")
    globs = [("www.alpha-evil.com", "glob 0"), ("beta-site.net", "globs 1 and 10"), ("cdn7.gamma.org", "glob 2"), ("cdnx.gamma.org", "class mismatch"),
             ("delta9.io", "glob 3"), ("delta.io", "? needs one character"), ("an-epsilon-name.example", "glob 4"), ("a.b.foxtrot.co.uk", "glob 5"),
             ("golf.x.com", "glob 6"), ("hotel-zulu.org", "glob 7"), ("hotel-alpha.org", "negated class"), ("india.example.com", "literal pattern 8"),
             ("x.india.example.com.y", "literal patterns match as substrings (Q9)"), ("www.alpha-evil.co", "glob 9 (its AC literal fails over into the branch of glob 0's)"),
             ("bert.net", "pure wildcard glob 10"), ("dexter-lab.biz", "glob 11 through the SPARSE node"), ("dexter-lab.com", "literal found, glob fails"),
             ("nothing.example", "no pattern")]
    for q, why in globs:
        ids = []
        for pid, (pat, ptype, lit) in enumerate(patterns):
            if (ptype == 0 and pat in q) or (ptype == 1 and fnmatch.fnmatchcase(q, pat)):
                ids.append(pid)
        e = {"kind": "pattern", "ids": ids, "data": [{"g": i} for i in ids]} if ids else {"kind": "notfound"}
        add(q, e, why)
    return ex


def main():
    out = {}
    for name, rs, ipv in (("24", 24, 4), ("28", 28, 4), ("32", 32, 4), ("v6", 24, 6)):
        blob, patterns, literals = assemble(rs, ipv)
        (HERE / f"handmade_{name}.mxy").write_bytes(blob)
        out[name] = {"record_size": rs, "ip_version": ipv, "bytes": len(blob), "queries": expectations(ipv, patterns)}
    (HERE / "handmade_expect.json").write_text(json.dumps(out, indent=1, ensure_ascii=False) + "\n")
    for k, v in out.items():
        print(k, v["bytes"], "bytes,", len(v["queries"]), "queries")


if __name__ == "__main__":
    main()
