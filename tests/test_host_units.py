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


def test_mutated_database_files_under_sanitizers(tmp_path):
    # The host side of the .mxy reader under AddressSanitizer + UBSan (GPU sanitizers are not available on the pool): mutated
    # copies of builder-made and handmade databases are either rejected with a message or opened and then walked the way the
    # device upload walks them (tree nodes, literal table, AC literal map, flattened automaton, pattern strings, data values)
    # without an out-of-bounds access or an allocation sized by a field of the file.
    import os
    csrc = ROOT / "matchy_amd" / "csrc"
    exe = tmp_path / "fuzz_db_image"
    subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                    "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I", str(csrc), str(ROOT / "tests/cpp/fuzz_db_image.cpp"),
                    *(str(csrc / f) for f in ("db_image.cpp", "data_codec.cpp", "db_builder.cpp", "unicode_lower.cpp", "host_lookup.cpp")), "-o", str(exe)], check=True)
    seeds = sorted(str(p) for p in (ROOT / "tests" / "golden").glob("handmade_*.mxy"))
    assert seeds
    env = dict(os.environ, MATCHY_AMD_LOWERCASE=str(ROOT / "matchy_amd" / "data" / "lowercase.bin"))
    r = subprocess.run([str(exe), "4000", "7", *seeds], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert r.stdout.startswith("OK: 4000 mutated images")


def test_xxh64_every_length_against_the_xxhash_module(tmp_path):
    # hashes.h reads short keys and the last bytes of a key through overlapping 8-byte loads: every length 0..260, plain and with
    # the ASCII case fold, against an implementation this repo did not write
    import xxhash
    out = _build_and_run("tests/cpp/dump_xxh64.cpp", tmp_path)
    data = bytes((i * 131 + 7) & 0xFF for i in range(300))
    lower = lambda x: bytes(c + 32 if 65 <= c <= 90 else c for c in x)
    rows = [line.split() for line in out.splitlines()]
    assert len(rows) == 261
    for n, h, hf in rows:
        n = int(n)
        assert int(h, 16) == xxhash.xxh64(data[:n], seed=0).intdigest(), n
        assert int(hf, 16) == xxhash.xxh64(lower(data[:n]), seed=0).intdigest(), n


def test_numa_mapping_of_a_gpu_to_its_cpus(tmp_path):
    """hipDeviceGetPCIBusId -> numa_node -> cpulist: the mapping the workers of the multi-device scanner and the ranks of bench.py
    bind their threads with, over a made-up sysfs tree (no GPU, no HIP call)."""
    import matchy_amd as M
    root = tmp_path / "sys"
    for bus, node in (("0000:c1:00.0", "1"), ("0000:05:00.0", "0"), ("0000:e9:00.0", "-1")):
        d = root / "bus" / "pci" / "devices" / bus
        d.mkdir(parents=True)
        (d / "numa_node").write_text(node + "\n")
    for node, cpus in ((0, "0-3,96-99\n"), (1, "48-50, 144,146-147\n")):
        d = root / "devices" / "system" / "node" / f"node{node}"
        d.mkdir(parents=True)
        (d / "cpulist").write_text(cpus)
    assert M.numa_cpus(str(root), "0000:05:00.0") == [0, 1, 2, 3, 96, 97, 98, 99]
    assert M.numa_cpus(str(root), "0000:C1:00.0") == [48, 49, 50, 144, 146, 147]   # HIP prints the bus id in either case
    assert M.numa_cpus(str(root), "0000:e9:00.0") == []    # the platform does not say: the thread is left alone
    assert M.numa_cpus(str(root), "0000:77:00.0") == []    # unknown device
    # malformed lists end where they stop making sense
    (root / "devices" / "system" / "node" / "node0" / "cpulist").write_text("5-2,9\n")
    assert M.numa_cpus(str(root), "0000:05:00.0") == []
    (root / "devices" / "system" / "node" / "node0" / "cpulist").write_text("1,3-4,x\n")
    assert M.numa_cpus(str(root), "0000:05:00.0") == [1, 3, 4]


def _host_lookup_exe(tmp_path_factory_dir):
    import os
    csrc = ROOT / "matchy_amd" / "csrc"
    exe = tmp_path_factory_dir / "host_lookup_cli"
    if not exe.exists():
        subprocess.run(["g++", "-O1", "-g", "-std=c++17", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                        "-D__HIP_PLATFORM_AMD__", "-I", "/opt/rocm/include", "-I", str(csrc), str(ROOT / "tests/cpp/host_lookup_cli.cpp"),
                        *(str(csrc / f) for f in ("host_lookup.cpp", "db_image.cpp", "data_codec.cpp", "unicode_lower.cpp")), "-o", str(exe)], check=True)
    return exe


def _host_answers(exe, blob, queries, tmp_path):
    import json
    import os
    (tmp_path / "db.mxy").write_bytes(blob)
    (tmp_path / "q.txt").write_bytes(b"".join(q.encode("utf-8") + b"\n" for q in queries))
    env = dict(os.environ, MATCHY_AMD_LOWERCASE=str(ROOT / "matchy_amd" / "data" / "lowercase.bin"))
    r = subprocess.run([str(exe), str(tmp_path / "db.mxy"), str(tmp_path / "q.txt")], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    out = [json.loads(l) for l in r.stdout.splitlines()]
    assert len(out) == len(queries)
    return out


def test_host_query_path_against_the_oracle(tmp_path):
    """csrc/host_lookup.cpp — what answers matchy_query since round 5 — compiled on its own with g++ under AddressSanitizer + UBSan (no GPU, no
    library) against the oracle's Database::lookup: the handmade files (24 / 28 / 32-bit trees, IPv4 in an IPv6 tree, EMPTY / ONE / SPARSE / DENSE
    automaton nodes, a full ACLH table, a pure wildcard), the tree.rs record vectors, builder-made databases of the synthetic configurations,
    a glob fuzz case (stars, question marks, classes, multi-byte characters, near misses) and a case-insensitive database."""
    import json
    import random
    import sys
    sys.path.insert(0, str(ROOT))
    import matchy_amd as M
    from oracle import oracle
    from tools import synth
    from tests.test_builder_oracle import _tree_kat
    oracle.build()
    exe = _host_lookup_exe(tmp_path)
    gold = ROOT / "tests" / "golden"
    cases = []
    exp = json.loads((gold / "handmade_expect.json").read_text())
    for name in ("24", "28", "32", "v6"):
        cases.append((f"handmade_{name}", (gold / f"handmade_{name}.mxy").read_bytes(), [q["query"] for q in exp[name]["queries"]]))
    for name, (blob, _, queries) in _tree_kat().items():
        if len(blob) < (1 << 22):
            cases.append((name, blob, [q for q, _ in queries]))
    rng = random.Random(3)
    for cfgname in ("c1", "c4/50", "c3b/200"):
        cfg = synth.config(cfgname)
        keys = [k.decode() for k, _ in synth.ioc_entries(cfg)]
        qs = rng.sample(keys, min(len(keys), 300))
        for k in list(qs):
            if "/" in k and ":" not in k:
                qs.append(k.split("/")[0])
            if k.startswith("*."):
                qs += ["www" + k[1:], "a.b" + k[1:], k[2:]]
            if k.startswith("glob:"):
                qs += [k[5:], "x" + k[5:] + "y"]
        import re
        toks = re.findall(rb"[0-9A-Za-z][0-9A-Za-z.:\-]{3,80}", synth.make_log(cfg, 0, 1500))
        qs += [t.decode() for t in rng.sample(toks, 400)]
        qs += ["", ".", "1.2.3.4", "::", "::1", "2001:db8::1", "::ffff:1.2.3.4", "münchen.de", "x" * 300, "EXAMPLE.COM"]
        cases.append((cfgname, synth.build_db(cfg), qs))
    # glob fuzz (the generator of the GPU glob test) and a case-insensitive twin of it
    sys.path.insert(0, str(ROOT / "tests"))
    from tests.test_gpu_parity import _glob_fuzz_case
    pats, log = _glob_fuzz_case(11)
    names = [t.decode("utf-8") for t in re.split(rb"[ \n/\"]+", log) if t]
    for ci in (False, True):
        b = M.DatabaseBuilder(build_epoch=6, case_insensitive=ci)
        for pt, i in pats.items():
            b.add_entry(pt, {"g": i})
        b.add_entry("literal:" + names[0], {"lit": True})
        b.add_entry("Straße.Example", {"s": 1})
        qs = names[:500] + ([n.upper() for n in names[:200]] if ci else []) + ["STRASSE.EXAMPLE", "straße.example", "Straße.Example"]
        cases.append((f"globfuzz ci={ci}", b.build(), qs))
    for name, blob, qs in cases:
        qs = [q for q in qs if "\n" not in q]
        got = _host_answers(exe, blob, qs, tmp_path)
        odb = oracle.Database(blob)
        n_found = 0
        for q, g in zip(qs, got):
            want = odb.lookup(q)
            if want["kind"] in ("notfound", "none"):
                assert g == {"kind": "notfound"}, (name, q, g)
            else:
                n_found += 1
                assert g == want, (name, q, g, want)
        assert n_found >= 3, name


def test_host_query_path_on_the_glob_vectors_of_the_reference(tmp_path):
    """the same vectors (glob.rs:464-705, tests/golden/glob_kat.json) through csrc/host_lookup.cpp under ASan / UBSan"""
    import sys
    sys.path.insert(0, str(ROOT))
    from tests.test_builder_oracle import _glob_kat_dbs
    exe = _host_lookup_exe(tmp_path)
    for c, blob in _glob_kat_dbs():
        qs = [t for t in c["match"] + c["nomatch"] if "\n" not in t]
        got = _host_answers(exe, blob, qs, tmp_path)
        for t, g in zip(qs, got):
            assert (g["kind"] == "pattern") == (t in c["match"]), (c["ref"], c["pattern"], t, g)


def test_host_query_path_on_the_exact_ip_vectors_of_the_reference(tmp_path):
    """test_ip_exact_match.rs through csrc/host_lookup.cpp (ASan / UBSan)"""
    import sys
    sys.path.insert(0, str(ROOT))
    from tests.test_builder_oracle import IP_EXACT_MATCH_KAT, build
    exe = _host_lookup_exe(tmp_path)
    for ref, entries, found, missing in IP_EXACT_MATCH_KAT:
        got = _host_answers(exe, build(entries), found + missing, tmp_path)
        for q, g in zip(found + missing, got):
            assert (g["kind"] == "ip") == (q in found), (ref, q, g)


def test_host_query_path_on_the_paraglob_vectors_of_the_reference(tmp_path):
    """matchy-paraglob/tests/integration_tests.rs:11-260 through csrc/host_lookup.cpp (ASan / UBSan)"""
    import sys
    sys.path.insert(0, str(ROOT))
    from tests.test_builder_oracle import PARAGLOB_KAT, paraglob_kat_check, paraglob_kat_db
    exe = _host_lookup_exe(tmp_path)
    for ref, ci, patterns, checks in PARAGLOB_KAT:
        got = _host_answers(exe, paraglob_kat_db(patterns, ci), [t for t, _ in checks], tmp_path)
        for (text, expect), g in zip(checks, got):
            ids = g.get("pattern_ids", []) if g["kind"] == "pattern" else []
            assert paraglob_kat_check(ids, expect), (ref, text, expect, g)


def test_host_query_path_on_the_literal_hash_vectors_of_the_reference(tmp_path):
    """crates/matchy/tests/test_literal_hash.rs:52-265 through csrc/host_lookup.cpp (ASan / UBSan)"""
    import sys
    sys.path.insert(0, str(ROOT))
    from tests.test_builder_oracle import LITERAL_HASH_KAT, build
    exe = _host_lookup_exe(tmp_path)
    for ref, entries, checks in LITERAL_HASH_KAT:
        if not checks:
            continue
        got = _host_answers(exe, build(entries), [q for q, _, _ in checks], tmp_path)
        for (q, kind, n), g in zip(checks, got):
            assert g["kind"] == kind and len(g.get("pattern_ids", [])) == n, (ref, q, g)
