#!/usr/bin/env python3
"""IP-tree record vectors the reference's own tests hold (crates/matchy-format/src/mmdb/tree.rs:322-398), as whole `.mxy`
files so that they go through the readers, the uploader and the lookup kernels:

  * test_read_24bit_record: node 0 = bytes 00 00 01 | 00 00 02, node_count 10  -> left record 1, right record 2
  * test_read_28bit_record: node 0 = bytes 00 00 01 | 12 | 00 00 02, node_count 10 -> left 0x1000001, right 0x2000002
    (the middle byte carries the HIGH nibble of the left record in its high half and of the right record in its low half —
    no file the handmade set of make_handmade_mxy.py contains has a non-zero nibble there)
  * test_calculate_data_offset: node_count 100: record 116 -> data offset 0, record 200 -> data offset 84

A record above node_count is a data pointer: data offset = record - node_count - 16 (tree.rs:218-245). With node_count 10
the 28-bit vector's records therefore point 16 777 191 and 33 554 408 bytes into the data section: the files are ~32 MiB,
so they are ASSEMBLED AT TEST TIME (files() below), not committed. The same two records are also written with 32-bit
records (big-endian words 01 00 00 01 | 02 00 00 02).

Expected answers follow from the construction (an IPv4 tree: the first address bit picks the record of node 0):
    kat28 / kat32:  0.0.0.0/1  -> record 0x1000001 -> data {"side": "left"},  prefix_len 1
                    128.0.0.0/1 -> record 0x2000002 -> data {"side": "right"}, prefix_len 1
    kat24:          node 0 -> (node 1, node 2); node 1 = (data A, not found); node 2 = (not found, data B):
                    0.0.0.0/2 -> A, 64.0.0.0/2 -> none, 128.0.0.0/2 -> none, 192.0.0.0/2 -> B, prefix_len 2
    kat_off:        node_count 100: node 0 = (record 116 -> offset 0, node 1); node 1 = (record 200 -> offset 84, not found)
Nodes no walk reaches hold "not found" records (= node_count) like the empty slots of a real tree.
"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent))
from make_handmade_mxy import U16, U32, U64, enc  # noqa: E402


def _meta(record_size, node_count, n_entries):
    return b"\xAB\xCD\xEFMaxMind.com" + enc({
        "binary_format_major_version": U16(2), "binary_format_minor_version": U16(0), "build_epoch": U64(1700000000),
        "database_type": "Tree-KAT", "description": {"en": "tests/golden/make_tree_kat.py"}, "languages": ["en"],
        "ip_version": U16(4), "node_count": U32(node_count), "record_size": U16(record_size), "ip_entry_count": U32(n_entries),
        "literal_entry_count": U32(0), "glob_entry_count": U32(0), "match_mode": U16(0),
        "pattern_section_offset": U32(0), "literal_section_offset": U32(0)})


def _node(record_size, left, right):
    if record_size == 24:
        return left.to_bytes(3, "big") + right.to_bytes(3, "big")
    if record_size == 28:
        return (left & 0xFFFFFF).to_bytes(3, "big") + bytes([((left >> 24) << 4) | (right >> 24)]) + (right & 0xFFFFFF).to_bytes(3, "big")
    return left.to_bytes(4, "big") + right.to_bytes(4, "big")


def _file(record_size, node_count, nodes, data_at):
    """nodes: {index: (left, right)}; data_at: {data offset: value}. Unlisted nodes are (node_count, node_count)."""
    tree = b"".join(_node(record_size, *nodes.get(i, (node_count, node_count))) for i in range(node_count))
    size = max(off + len(enc(v)) for off, v in data_at.items())
    data = bytearray(size)
    for off, v in data_at.items():
        b = enc(v)
        data[off:off + len(b)] = b
    return tree + bytes(16) + bytes(data) + _meta(record_size, node_count, len(data_at))


def files():
    """-> {name: (blob, node 0 bytes the reference test writes, [(query, expected)])}; expected = (prefix_len, data) or None."""
    out = {}
    nc = 10
    left, right = 0x1000001, 0x2000002
    for name, rs, node0 in (("kat28", 28, bytes([0, 0, 1, 0x12, 0, 0, 2])), ("kat32", 32, bytes([1, 0, 0, 1, 2, 0, 0, 2]))):
        blob = _file(rs, nc, {0: (left, right)}, {left - nc - 16: {"side": "left"}, right - nc - 16: {"side": "right"}})
        q = [("0.0.0.0", (1, {"side": "left"})), ("1.2.3.4", (1, {"side": "left"})), ("127.255.255.255", (1, {"side": "left"})),
             ("128.0.0.0", (1, {"side": "right"})), ("200.1.1.1", (1, {"side": "right"})), ("255.255.255.255", (1, {"side": "right"}))]
        out[name] = (blob, node0, q)
    a_off, b_off = 0, 40
    blob = _file(24, nc, {0: (1, 2), 1: (nc + 16 + a_off, nc), 2: (nc, nc + 16 + b_off)}, {a_off: {"leaf": "A"}, b_off: {"leaf": "B"}})
    out["kat24"] = (blob, bytes([0, 0, 1, 0, 0, 2]),
                    [("10.0.0.1", (2, {"leaf": "A"})), ("64.0.0.1", None), ("100.64.0.1", None), ("128.0.0.1", None), ("191.255.0.1", None),
                     ("192.0.2.1", (2, {"leaf": "B"})), ("255.0.0.0", (2, {"leaf": "B"}))])
    nc = 100
    blob = _file(24, nc, {0: (116, 1), 1: (200, nc)}, {0: {"off": U16(0)}, 84: {"off": U16(84)}})
    out["kat_off"] = (blob, (116).to_bytes(3, "big") + (1).to_bytes(3, "big"),
                      [("1.2.3.4", (1, {"off": 0})), ("128.0.0.1", (2, {"off": 84})), ("191.1.1.1", (2, {"off": 84})), ("192.0.0.1", None)])
    for name, (blob, node0, _) in out.items():
        assert blob[:len(node0)] == node0, name
    return out


if __name__ == "__main__":
    for name, (blob, node0, q) in files().items():
        print(name, len(blob), "bytes, node 0 =", node0.hex(), ",", len(q), "queries")
