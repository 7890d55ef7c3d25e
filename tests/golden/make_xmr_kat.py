#!/usr/bin/env python3
"""Writes tests/golden/xmr_kat.json: checksum-VALID Monero vectors for E9.

The reference's own Monero test (crates/matchy-extractor/src/lib.rs:3449-3476) asserts nothing when the address is not
extracted, and its two other Monero tests are rejects, so the accept branch of `validate_monero_address`
(lib.rs:1895-1920) is pinned by no vector the reference holds. The rule itself is fully specified there:

  * token between word boundaries, 90..=110 bytes, first byte '4' or '8'            (lib.rs:1383-1397)
  * `bs58::decode(addr)`: WHOLE-STRING Base58 over the Bitcoin alphabet (a big-endian bignum; each leading '1' is one
    leading zero byte — impossible here because the first byte is '4' or '8')       (lib.rs:1899-1903)
  * decoded length >= 5; last 4 bytes == first 4 bytes of Keccak-256 (tiny-keccak `Keccak::v256`: the ORIGINAL Keccak
    padding 0x01, not SHA-3's 0x06) of everything before them                        (lib.rs:1905-1919)

So accepts can be CONSTRUCTED: pick a bignum whose Base58 text has the wanted length and first digit, take all but its
last four bytes as the payload, replace the last four with the checksum (the low 32 bits cannot move the first digit
or the length except with probability ~2^-400) and encode. Everything below is plain Python integer arithmetic; the
Keccak-f[1600] permutation is written from the Keccak reference (FIPS 202 §3.2 step mappings) and pinned in this
script by the Keccak-team vectors for Keccak-256("") and Keccak-256("abc") — the same two vectors
tests/test_oracle_kat.py::test_primitives_kat holds against the oracle.

(Real-world Monero addresses use a BLOCK-wise Base58 — 8 bytes -> 11 characters — so the reference's whole-string
decode rejects nearly all of them; that is the reference's behaviour and these vectors follow the reference, not
Monero.)

Nothing here imports the oracle or the product: the expected outcomes follow from the construction.
"""
import json
import random
from pathlib import Path

ALPHABET = "123456789ABCDEFGHJKLMNPQRSTUVWXYZabcdefghijkmnopqrstuvwxyz"
M64 = (1 << 64) - 1
RC = [0x0000000000000001, 0x0000000000008082, 0x800000000000808A, 0x8000000080008000, 0x000000000000808B, 0x0000000080000001,
      0x8000000080008081, 0x8000000000008009, 0x000000000000008A, 0x0000000000000088, 0x0000000080008009, 0x000000008000000A,
      0x000000008000808B, 0x800000000000008B, 0x8000000000008089, 0x8000000000008003, 0x8000000000008002, 0x8000000000000080,
      0x000000000000800A, 0x800000008000000A, 0x8000000080008081, 0x8000000000008080, 0x0000000080000001, 0x8000000080008008]
ROT = [[0, 36, 3, 41, 18], [1, 44, 10, 45, 2], [62, 6, 43, 15, 61], [28, 55, 25, 21, 56], [27, 20, 39, 8, 14]]  # [x][y]


def _rol(v, n):
    n %= 64
    return ((v << n) | (v >> (64 - n))) & M64 if n else v


def keccak_f(a):  # a[x][y]
    for rc in RC:
        c = [a[x][0] ^ a[x][1] ^ a[x][2] ^ a[x][3] ^ a[x][4] for x in range(5)]
        d = [c[(x - 1) % 5] ^ _rol(c[(x + 1) % 5], 1) for x in range(5)]
        a = [[a[x][y] ^ d[x] for y in range(5)] for x in range(5)]
        b = [[0] * 5 for _ in range(5)]
        for x in range(5):
            for y in range(5):
                b[y][(2 * x + 3 * y) % 5] = _rol(a[x][y], ROT[x][y])
        a = [[b[x][y] ^ ((~b[(x + 1) % 5][y]) & b[(x + 2) % 5][y]) for y in range(5)] for x in range(5)]
        a[0][0] ^= rc
    return a


def keccak256(msg: bytes) -> bytes:
    rate = 136
    p = bytearray(msg) + b"\x01"          # original Keccak multi-rate padding (tiny-keccak Keccak::v256)
    p += b"\x00" * (-len(p) % rate)
    p[-1] |= 0x80
    a = [[0] * 5 for _ in range(5)]
    for off in range(0, len(p), rate):
        for i in range(rate // 8):
            a[i % 5][i // 5] ^= int.from_bytes(p[off + 8 * i: off + 8 * i + 8], "little")
        a = keccak_f(a)
    return b"".join(a[i % 5][i // 5].to_bytes(8, "little") for i in range(4))


assert keccak256(b"").hex() == "c5d2460186f7233c927e7db2dcc703c0e500b653ca82273b7bfad8045d85a470"
assert keccak256(b"abc").hex() == "4e03657aea45a94fc7d47ba826c8d667c0d1e6e33a64a036ec44f58fa12d6c45"
assert keccak256(b"a" * 200).hex() != keccak256(b"a" * 199).hex()      # two-block absorb runs


def b58encode(b: bytes) -> str:
    v = int.from_bytes(b, "big")
    s = ""
    while v:
        v, r = divmod(v, 58)
        s = ALPHABET[r] + s
    return "1" * (len(b) - len(b.lstrip(b"\x00"))) + s


def b58decode(s: str) -> bytes:
    v = 0
    for ch in s:
        v = v * 58 + ALPHABET.index(ch)
    z = len(s) - len(s.lstrip("1"))
    return b"\x00" * z + (v.to_bytes((v.bit_length() + 7) // 8, "big") if v else b"")


assert b58encode(bytes.fromhex("00010966776006953D5567439E5E39F86A0D273BEEd61967f6")) == "16UwLL9Risc3QfPqBUvKofHmBQ7wMtjvM"
assert b58decode("16UwLL9Risc3QfPqBUvKofHmBQ7wMtjvM").hex() == "00010966776006953d5567439e5e39f86a0d273beed61967f6"


def make_accept(rng, length, first):
    """A checksum-valid token of exactly `length` Base58 characters whose first character is `first`."""
    lo = ALPHABET.index(first) * 58 ** (length - 1)
    while True:
        v = lo + rng.randrange(58 ** (length - 1))
        raw = v.to_bytes((v.bit_length() + 7) // 8, "big")
        raw = raw[:-4] + keccak256(raw[:-4])[:4]
        s = b58encode(raw)
        if len(s) == length and s[0] == first:
            assert b58decode(s) == raw and keccak256(raw[:-4])[:4] == raw[-4:]
            return s


def flip(s, i):
    """Another alphabet character at position i (never a boundary byte, never changes the length)."""
    c = ALPHABET[(ALPHABET.index(s[i]) + 1) % 58] if i else s[i]
    return s[:i] + c + s[i + 1:]


def main():
    rng = random.Random(0x786D72)
    accepts, rejects = [], []
    for length in (95, 106, 90, 110, 100):
        for first in "48":
            for _ in range(2 if length in (95, 106) else 1):
                a = make_accept(rng, length, first)
                accepts.append(a)
                rejects.append({"text": flip(a, rng.randrange(1, length - 1)), "why": "one character changed"})
                rejects.append({"text": flip(a, length - 1), "why": "last character changed"})
    a95 = accepts[0]
    # the gates in front of the checksum (lib.rs:1386-1397) on otherwise valid text, and bs58's alphabet errors
    rejects.append({"text": make_accept(rng, 89, "4"), "why": "89 characters: checksum fine, too short"})
    rejects.append({"text": make_accept(rng, 111, "8"), "why": "111 characters: checksum fine, too long"})
    rejects.append({"text": make_accept(rng, 95, "5"), "why": "first character 5: checksum fine, wrong prefix"})
    rejects.append({"text": make_accept(rng, 95, "9"), "why": "first character 9: checksum fine, wrong prefix"})
    for bad in "0OIl":
        rejects.append({"text": a95[:40] + bad + a95[41:], "why": f"'{bad}' is not in the Base58 alphabet"})
    out = Path(__file__).with_name("xmr_kat.json")
    out.write_text(json.dumps({"ref": "crates/matchy-extractor/src/lib.rs:1367-1409,1895-1920", "accept": accepts, "reject": rejects},
                              indent=1) + "\n")
    print(len(accepts), "accepts,", len(rejects), "rejects ->", out)


if __name__ == "__main__":
    main()
