#!/usr/bin/env python3
"""Generate matchy_amd/data/psl.bin from a Public Suffix List snapshot.

The reference embeds `crates/matchy-extractor/src/data/public_suffix_list.dat`
(include_str!, matchy-extractor/src/lib.rs:1546) and builds a byte-string set
from its non-empty, non-`//` lines after `trim()` (lib.rs:1552-1563).  We apply
exactly that filter to the same public data snapshot and store the resulting
set in our own compact container so that the oracle and the device-table
builder both consume identical suffix sets on a box without /root/reference.

Container "PSLB" v1 (little endian):
    magic[4]="PSLB"  version u32  count u32  payload_bytes u32
    then `count` front-coded entries, sorted bytewise ascending:
        u8 shared   (bytes shared with the previous entry)
        u8 rest     (bytes that follow)
        rest bytes
"""
import struct
import sys
from pathlib import Path

DEFAULT_SRC = Path("/root/reference/crates/matchy-extractor/src/data/public_suffix_list.dat")
DST = Path(__file__).resolve().parent.parent / "matchy_amd" / "data" / "psl.bin"


def load_suffixes(path: Path):
    out = set()
    for line in path.read_text(encoding="utf-8").splitlines():
        line = line.strip()
        if not line or line.startswith("//"):
            continue
        out.add(line.encode("utf-8"))
    return sorted(out)


def encode(suffixes):
    payload = bytearray()
    prev = b""
    for s in suffixes:
        shared = 0
        m = min(len(prev), len(s), 255)
        while shared < m and prev[shared] == s[shared]:
            shared += 1
        rest = s[shared:]
        if len(rest) > 255:
            raise ValueError("suffix too long: %r" % s)
        payload += bytes((shared, len(rest))) + rest
        prev = s
    return struct.pack("<4sIII", b"PSLB", 1, len(suffixes), len(payload)) + bytes(payload)


def main():
    src = Path(sys.argv[1]) if len(sys.argv) > 1 else DEFAULT_SRC
    suffixes = load_suffixes(src)
    blob = encode(suffixes)
    DST.parent.mkdir(parents=True, exist_ok=True)
    DST.write_bytes(blob)
    print(f"{len(suffixes)} suffixes -> {DST} ({len(blob)} bytes)")


if __name__ == "__main__":
    main()
