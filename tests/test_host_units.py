"""CPU: host-side unit checks of code the HIP kernels share with the host (compiled with g++, no GPU)."""
import subprocess
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def _build_and_run(src, tmp_path, extra=()):
    exe = tmp_path / Path(src).stem
    subprocess.run(["g++", "-O1", "-std=c++17", "-I", str(ROOT / "matchy_amd" / "csrc"), *extra, str(ROOT / src), "-o", str(exe)], check=True)
    return subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout


def test_anchor_bit_planes(tmp_path):
    # bit transpose + byte classes of k_anchor's bit-sliced front end, all byte values in every bit slot
    out = _build_and_run("tests/cpp/test_anchor_planes.cpp", tmp_path)
    assert "anchor_planes ok" in out


def test_xxh64_stream(tmp_path):
    # the byte-at-a-time XXH64 used for lower-cased non-ASCII queries equals the one-shot hash (spec vector included)
    out = _build_and_run("tests/cpp/test_xxh64_stream.cpp", tmp_path)
    assert "xxh64 stream ok" in out
