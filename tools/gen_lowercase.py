"""Writes matchy_amd/data/lowercase.bin: the Unicode default lower-case mapping that Rust's `str::to_lowercase` applies
(used by the reference for case-insensitive databases: matchy-literal-hash/src/lib.rs:162-165,469-472, matchy-ac/src/lib.rs:209).

Data, not code. Source of the data: this interpreter's Unicode tables (`str.lower()`, unicodedata.unidata_version is written
into the header) — the reference's toolchain is unpinned, so the Unicode version is not pinned either; characters added after
this version would map to themselves here.

Layout (little-endian): "LCTB", u32 version (major << 16 | minor << 8 | patch), u32 n_map, u32 n_ign, u32 n_cased,
  n_map   x {u32 code point, u8 len, u8 utf8[7]}   code points (sorted) whose lower-case form differs, with that form's UTF-8
  n_ign   x {u32 first, u32 last}                  Case_Ignorable ranges   } the Final_Sigma rule for U+03A3
  n_cased x {u32 first, u32 last}                  Cased ranges            } (Rust: alloc/src/str.rs map_uppercase_sigma)
The two property sets are recovered from the interpreter's own Final_Sigma handling: U+03A3 lower-cases to U+03C2 iff it is
preceded by a cased character (skipping case-ignorable ones) and not followed by one.
"""
import struct
import sys
import unicodedata
from pathlib import Path

OUT = Path(__file__).resolve().parent.parent / "matchy_amd" / "data" / "lowercase.bin"


def ranges(flags):
    out, start = [], None
    for cp, f in enumerate(flags):
        if f and start is None:
            start = cp
        if not f and start is not None:
            out.append((start, cp - 1))
            start = None
    if start is not None:
        out.append((start, len(flags) - 1))
    return out


def main():
    maps = []
    ign = [False] * 0x110000
    cased = [False] * 0x110000
    for cp in range(0x110000):
        if 0xD800 <= cp <= 0xDFFF:
            continue
        ch = chr(cp)
        lo = ch.lower()
        if lo != ch and cp != 0x3A3:
            b = lo.encode("utf-8")
            assert len(b) <= 7
            maps.append((cp, b))
        # Final_Sigma probes (see the module docstring)
        t1 = (ch + "Σ").lower().endswith("ς")          # not ignorable and cased
        t2 = ("A" + ch + "Σ").lower().endswith("ς")    # ignorable, or (not ignorable and cased)
        if t1:
            cased[cp] = True
        elif t2:
            ign[cp] = True
    assert all(cp >= 0x41 for cp, _ in maps)
    ri, rc = ranges(ign), ranges(cased)
    ver = [int(x) for x in unicodedata.unidata_version.split(".")]
    blob = bytearray(b"LCTB")
    blob += struct.pack("<IIII", (ver[0] << 16) | (ver[1] << 8) | ver[2], len(maps), len(ri), len(rc))
    for cp, b in maps:
        blob += struct.pack("<IB7s", cp, len(b), b)
    for a, z in ri + rc:
        blob += struct.pack("<II", a, z)
    OUT.write_bytes(blob)
    print(f"{OUT}: Unicode {unicodedata.unidata_version}, {len(maps)} mappings, {len(ri)} case-ignorable ranges, {len(rc)} cased ranges, {len(blob)} B")


if __name__ == "__main__":
    sys.exit(main())
