"""Writes matchy_amd/data/lowercase.bin: the Unicode default lower-case mapping that Rust's `str::to_lowercase` applies
(used by the reference for case-insensitive databases: matchy-literal-hash/src/lib.rs:162-165,469-472, matchy-ac/src/lib.rs:209).

Data, not code. Source of the data: this interpreter's Unicode tables (`str.lower()`; Unicode 13 in this image) PLUS the case
pairs Unicode 14.0 and 16.0 added (15.0 / 15.1 added none), hand-listed below from the standard's UnicodeData.txt — a current
Rust toolchain lower-cases with Unicode 16 data. The reference's toolchain is unpinned, so this is "parity unpinned" either
way; tests/test_gpu_parity.py holds KATs for the hand-listed pairs. The version in the header is the newest one covered.

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


# Simple lower-case mappings of the cased letters added after Unicode 13 (UnicodeData.txt field 13), as (first, last, delta) runs
# or single (capital, lower) pairs.
ADDED_PAIRS = [
    # Unicode 14.0
    (0x2C2F, 0x2C5F),                      # GLAGOLITIC CAPITAL LETTER CAUDATE CHU
    (0xA7C0, 0xA7C1),                      # LATIN CAPITAL LETTER OLD POLISH O
    (0xA7D0, 0xA7D1),                      # LATIN CAPITAL LETTER CLOSED INSULAR G
    (0xA7D6, 0xA7D7),                      # LATIN CAPITAL LETTER MIDDLE SCOTS S
    (0xA7D8, 0xA7D9),                      # LATIN CAPITAL LETTER SIGMOID S
    # Unicode 16.0
    (0x1C89, 0x1C8A),                      # CYRILLIC CAPITAL LETTER TJE
    (0xA7CB, 0x0264),                      # LATIN CAPITAL LETTER RAMS HORN -> U+0264
    (0xA7CC, 0xA7CD),                      # LATIN CAPITAL LETTER S WITH DIAGONAL STROKE
    (0xA7DA, 0xA7DB),                      # LATIN CAPITAL LETTER LAMBDA
    (0xA7DC, 0x019B),                      # LATIN CAPITAL LETTER LAMBDA WITH STROKE -> U+019B
]
ADDED_RUNS = [
    # Unicode 14.0: Vithkuqi (capital + 0x27)
    (0x10570, 0x1057A, 0x27), (0x1057C, 0x1058A, 0x27), (0x1058C, 0x10592, 0x27), (0x10594, 0x10595, 0x27),
    # Unicode 16.0: Garay (capital + 0x20)
    (0x10D50, 0x10D65, 0x20),
]
# Cased / Case_Ignorable additions that matter for the Final_Sigma context (letters only; the combining marks added since Unicode 13
# are not listed: a capital sigma next to one of those takes the non-final form here)
ADDED_CASED_ONLY = [(0xA7D3, 0xA7D3), (0xA7D5, 0xA7D5), (0x1DF00, 0x1DF09), (0x1DF0B, 0x1DF1E), (0x1DF25, 0x1DF2A)]
ADDED_IGNORABLE = [(0x10780, 0x10785), (0x10787, 0x107B0), (0x107B2, 0x107BA), (0x1E030, 0x1E06D), (0x1E08F, 0x1E08F)]


def added_pairs():
    out = list(ADDED_PAIRS)
    for a, z, d in ADDED_RUNS:
        out += [(cp, cp + d) for cp in range(a, z + 1)]
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
    ver = [int(x) for x in unicodedata.unidata_version.split(".")]
    if ver[0] < 16:
        have = {cp for cp, _ in maps}
        for up, lo in added_pairs():
            if ver[0] >= 14 and up not in (0x1C89, 0xA7CB, 0xA7CC, 0xA7DA, 0xA7DC) and not (0x10D50 <= up <= 0x10D65):
                continue   # the interpreter already has the 14.0 pairs
            assert up not in have and chr(up).lower() == chr(up), hex(up)
            maps.append((up, chr(lo).encode("utf-8")))
            cased[up] = True
            cased[lo] = True
        maps.sort()
        for a, z in ADDED_CASED_ONLY:
            for cp in range(a, z + 1):
                cased[cp] = True
        for a, z in ADDED_IGNORABLE:
            for cp in range(a, z + 1):
                ign[cp] = True
                cased[cp] = False
        ver = [16, 0, 0]
    ri, rc = ranges(ign), ranges(cased)
    blob = bytearray(b"LCTB")
    blob += struct.pack("<IIII", (ver[0] << 16) | (ver[1] << 8) | ver[2], len(maps), len(ri), len(rc))
    for cp, b in maps:
        blob += struct.pack("<IB7s", cp, len(b), b)
    for a, z in ri + rc:
        blob += struct.pack("<II", a, z)
    OUT.write_bytes(blob)
    print(f"{OUT}: Unicode {unicodedata.unidata_version} + hand-listed pairs up to {ver[0]}.{ver[1]}, {len(maps)} mappings, {len(ri)} case-ignorable ranges, {len(rc)} cased ranges, {len(blob)} B")


if __name__ == "__main__":
    sys.exit(main())
