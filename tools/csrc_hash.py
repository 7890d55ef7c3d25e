"""Fingerprint of the kernel sources (matchy_amd/csrc/*): profiles/*_traffic.json records it, bench.py refuses a traffic
file whose fingerprint differs from the tree it runs from (a stale profile must not pass as current evidence)."""
import hashlib
from pathlib import Path

CSRC = Path(__file__).resolve().parent.parent / "matchy_amd" / "csrc"


def csrc_sha() -> str:
    h = hashlib.sha256()
    for f in sorted(CSRC.glob("*")):
        if f.is_file():
            h.update(f.name.encode())
            h.update(f.read_bytes())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    print(csrc_sha())
