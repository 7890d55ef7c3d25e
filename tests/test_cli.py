"""`matchy` command line (matchy_amd/bin/matchy): `build` runs on the host only; `match` needs the GPU (-m gpu).
Mirrors the reference's CLI tests (crates/matchy/tests/cli_tests.rs:407-439, 690-1009): behaviour, not bytes of the
reference's files."""
import json
import os
import stat
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
CLI = ROOT / "matchy_amd" / "bin" / "matchy"


@pytest.fixture(scope="module")
def cli():
    import matchy_amd.build as B
    B.build()
    assert CLI.exists()
    return str(CLI)


def _run(args, **kw):
    return subprocess.run(args, capture_output=True, timeout=600, **kw)


def _c1_csv(path):
    from tools import synth
    cfg = synth.config("c1")
    rows = list(synth.ioc_entries(cfg))
    with open(path, "w") as f:
        f.write("entry,threat_level,category,source\n")
        for k, d in rows:
            d = json.loads(d)
            f.write(f"{k.decode()},{d['threat_level']},{d['category']},{d['source']}\n")
    return cfg, rows


def test_build_csv_matches_library_builder(cli, tmp_path):
    import matchy_amd as M
    cfg, rows = _c1_csv(tmp_path / "c1.csv")
    out = tmp_path / "c1.mxy"
    r = _run([cli, "build", str(tmp_path / "c1.csv"), "-o", str(out), "-f", "csv"], env=dict(os.environ, MATCHY_BUILD_EPOCH="7"))
    assert r.returncode == 0, r.stderr
    assert f"Database built: {out}" in r.stdout.decode()
    assert stat.S_IMODE(out.stat().st_mode) == 0o444  # build_cmd.rs:364-371
    b = M.DatabaseBuilder(build_epoch=7)
    for k, d in rows:
        b.add_entry(k.decode(), json.loads(d))
    assert out.read_bytes() == b.build()
    # rebuilding over the read-only file works
    r = _run([cli, "build", str(tmp_path / "c1.csv"), "-o", str(out), "-f", "csv"])
    assert r.returncode == 0, r.stderr


def test_build_csv_value_typing(cli, tmp_path, oracle):
    # build_cmd.rs:226-236: i64 -> Int32 (truncating), u64 -> Uint64, f64 -> Double, true/false -> Bool, empty dropped
    (tmp_path / "t.csv").write_text('key,n,big,huge,f,flag,s,empty,q\n1.2.3.4,42,4294967298,18446744073709551615,1.5,true,hello,,"a,b ""c"""\n')
    out = tmp_path / "t.mxy"
    assert _run([cli, "build", str(tmp_path / "t.csv"), "-o", str(out), "-f", "csv"]).returncode == 0
    db = oracle.Database(out.read_bytes())
    got = db.lookup("1.2.3.4")
    assert got["kind"] == "ip"
    assert got["data"] == {"n": 42, "big": 2, "huge": 18446744073709551615, "f": 1.5, "flag": True, "s": "hello", "q": 'a,b "c"'}


def test_build_text_and_json(cli, tmp_path, oracle):
    (tmp_path / "t.txt").write_text("# comment\n10.0.0.0/8\n\n  *.evil.com  \nexact.example.org\n")
    out = tmp_path / "t.mxy"
    assert _run([cli, "build", str(tmp_path / "t.txt"), "-o", str(out)]).returncode == 0
    db = oracle.Database(out.read_bytes())
    assert db.lookup("10.9.8.7")["kind"] == "ip"
    assert db.lookup("x.evil.com")["kind"] == "pattern"
    assert db.lookup("exact.example.org")["kind"] == "pattern"
    assert db.lookup("nope.example.org")["kind"] == "notfound"
    (tmp_path / "j.json").write_text(json.dumps([{"key": "8.8.8.8", "data": {"who": "dns", "n": 5}}, {"key": "*.bad.net"}]))
    out2 = tmp_path / "j.mxy"
    assert _run([cli, "build", str(tmp_path / "j.json"), "-o", str(out2), "-f", "json"]).returncode == 0
    db2 = oracle.Database(out2.read_bytes())
    assert db2.lookup("8.8.8.8")["data"] == {"who": "dns", "n": 5}
    assert db2.lookup("a.bad.net")["kind"] == "pattern"


def test_build_case_insensitive(cli, tmp_path, oracle):
    """cli_tests.rs:141-160 (`matchy build --case-insensitive`) + what the flag means for lookups."""
    (tmp_path / "patterns.txt").write_text("*.EVIL.COM\nBad.Example.ORG\n")
    out = tmp_path / "ci.mxy"
    r = _run([cli, "build", str(tmp_path / "patterns.txt"), "-o", str(out), "--case-insensitive"])
    assert r.returncode == 0 and out.exists(), r.stderr
    db = oracle.Database(out.read_bytes())
    assert db.metadata()["match_mode"] == 1
    assert db.lookup("www.evil.com")["kind"] == "pattern" and db.lookup("WWW.Evil.Com")["kind"] == "pattern"
    assert db.lookup("bad.example.org")["kind"] == "pattern" and db.lookup("BAD.EXAMPLE.ORG")["kind"] == "pattern"
    out2 = tmp_path / "cs.mxy"
    assert _run([cli, "build", str(tmp_path / "patterns.txt"), "-o", str(out2)]).returncode == 0
    db2 = oracle.Database(out2.read_bytes())
    assert db2.metadata()["match_mode"] == 0 and db2.lookup("www.evil.com")["kind"] != "pattern" and db2.lookup("www.EVIL.COM")["kind"] == "pattern"


def test_build_errors(cli, tmp_path):
    (tmp_path / "bad.csv").write_text("a,b\n1,2\n")
    r = _run([cli, "build", str(tmp_path / "bad.csv"), "-o", str(tmp_path / "x.mxy"), "-f", "csv"])
    assert r.returncode != 0 and b"'entry' or 'key' column" in r.stderr
    r = _run([cli, "build", str(tmp_path / "missing.txt"), "-o", str(tmp_path / "x.mxy")])
    assert r.returncode != 0
    assert _run([cli]).returncode == 2
    assert _run([cli, "build", str(tmp_path / "bad.csv")]).returncode == 2  # no -o


@pytest.mark.gpu
def test_match_config1_end_to_end(cli, tmp_path, oracle):
    """BASELINE configs[0]: 1K-indicator CSV -> .mxy, 10K-line access.log via `matchy match` (here: on the GPU)."""
    from tools import synth
    cfg, rows = _c1_csv(tmp_path / "c1.csv")
    dbp = tmp_path / "c1.mxy"
    assert _run([cli, "build", str(tmp_path / "c1.csv"), "-o", str(dbp), "-f", "csv"]).returncode == 0
    log = synth.make_log(cfg, 0, 10000)
    logp = tmp_path / "access.log"
    logp.write_bytes(log)
    want_hits, want_lines, st = oracle.Database(dbp.read_bytes()).scan(log, source=str(logp))
    r = _run([cli, "match", str(dbp), str(logp), "-s"])
    assert r.returncode == 0, r.stderr
    got = r.stdout.decode().splitlines()
    assert got == want_lines and len(got) > 50
    for line in got[:20]:
        obj = json.loads(line)
        assert list(obj) == sorted(obj) and obj["timestamp"] == "0.000" and obj["source"] == str(logp)
    err = r.stderr.decode()
    assert "[INFO] Lines processed: 10,000" in err and f"[INFO] Total matches: {len(got):,}" in err
    assert f"[INFO] Candidates tested: {st.candidates:,}" in err and "[INFO] Throughput:" in err
    # small batches (newline-aligned cuts + carry), stdin, CSV database built in memory: same match set
    r2 = _run([cli, "match", str(dbp), str(logp), "--batch-bytes", "65536"])
    assert r2.returncode == 0 and r2.stdout.decode().splitlines() == want_lines
    # the other input paths of regular files: byte ranges read with pread into pinned buffers; workers that fault and pin their
    # batches themselves (no reader-side feeding); a batch size above the pinning threshold so that the fed path really pins
    for env in ({"MATCHY_AMD_PREAD": "1"}, {"MATCHY_AMD_NO_FEEDER": "1"}, {"MATCHY_AMD_NO_REGISTER": "1"}):
        rp = _run([cli, "match", str(dbp), str(logp), "--batch-bytes", "65536", "-j", "3"], env=dict(os.environ, **env))
        assert rp.returncode == 0 and rp.stdout.decode().splitlines() == want_lines, env
    big = log * 6                                      # 10+ MB: batches of 5 MB are pinned by the reader
    bigp = tmp_path / "big.log"
    bigp.write_bytes(big)
    wb = [json.loads(l) for l in oracle.Database(dbp.read_bytes()).scan(big, source=str(bigp))[1]]
    for env in ({}, {"MATCHY_AMD_PREAD": "1"}, {"MATCHY_AMD_NO_FEEDER": "1"}):
        rb = _run([cli, "match", str(dbp), str(bigp), "--batch-bytes", str(5 << 20), "-j", "2"], env=dict(os.environ, **env))
        assert rb.returncode == 0 and [json.loads(l) for l in rb.stdout.decode().splitlines()] == wb, env
    r3 = _run([cli, "match", str(tmp_path / "c1.csv"), "-", "--batch-bytes=100000"], input=log)
    assert r3.returncode == 0
    assert [json.loads(l) | {"source": ""} for l in r3.stdout.decode().splitlines()] == [json.loads(l) | {"source": ""} for l in want_lines]
    # summary: statistics only; extractor selection; a missing input fails the run but the others are processed
    r4 = _run([cli, "match", str(dbp), str(logp), "--format", "summary", "-s"])
    assert r4.returncode == 0 and r4.stdout == b"" and b"Total matches" in r4.stderr
    r5 = _run([cli, "match", str(dbp), str(logp), "--extractors=ip"])
    ips = [json.loads(l) for l in r5.stdout.decode().splitlines()]
    assert ips and all(o["match_type"] == "ip" for o in ips)
    assert len(ips) == sum(1 for l in want_lines if json.loads(l)["match_type"] == "ip")
    # gzip input (by extension, file_reader.rs:45-75)
    import gzip
    gzp = tmp_path / "access.log.GZ"
    gzp.write_bytes(gzip.compress(log))
    rz = _run([cli, "match", str(dbp), str(gzp), "--batch-bytes", "300000"])
    assert rz.returncode == 0
    assert [json.loads(l) | {"source": ""} for l in rz.stdout.decode().splitlines()] == [json.loads(l) | {"source": ""} for l in want_lines]
    # several scanners (one worker per --devices entry; the same GPU twice here): batches are handed out in sequence and
    # printed in sequence, so output and counters do not depend on the device list
    rd = _run([cli, "match", str(dbp), str(logp), str(gzp), "--devices", "0,0,0", "--batch-bytes", "50000", "-s"])
    assert rd.returncode == 0, rd.stderr
    both = want_lines + [json.dumps(json.loads(l) | {"source": str(gzp)}, separators=(",", ":"), ensure_ascii=False) for l in want_lines]
    assert [json.loads(l) for l in rd.stdout.decode().splitlines()] == [json.loads(l) for l in both]
    assert b"[INFO] Lines processed: 20,000" in rd.stderr and b"3 scanners" in rd.stderr and b"[INFO] Files processed: 2" in rd.stderr
    # -j N|auto (the reference's worker-thread option): scanners per device list without repeats; auto = 4 per device when NDJSON is rendered
    rj = _run([cli, "match", str(dbp), str(logp), "-j", "3", "--batch-bytes", "50000", "-s"])
    assert rj.returncode == 0 and rj.stdout.decode().splitlines() == want_lines and b"3 scanners" in rj.stderr
    rj = _run([cli, "match", str(dbp), str(logp), "--batch-bytes", "50000", "-s"])
    assert rj.returncode == 0 and rj.stdout.decode().splitlines() == want_lines and b"4 scanners" in rj.stderr
    rj = _run([cli, "match", str(dbp), str(logp), "--batch-bytes", "50000", "-s", "--format", "summary"])
    assert rj.returncode == 0 and b"2 scanners" in rj.stderr and b"[INFO] Lines processed: 10,000" in rj.stderr
    ra = _run([cli, "match", str(dbp), str(logp), "--devices", "all"])
    assert ra.returncode == 0 and ra.stdout.decode().splitlines() == want_lines
    assert _run([cli, "match", str(dbp), str(logp), "--devices", "0,x"]).returncode == 1
    r6 = _run([cli, "match", str(dbp), str(tmp_path / "nope.log"), str(logp)])
    assert r6.returncode != 0 and r6.stdout.decode().splitlines() == want_lines


@pytest.mark.gpu
@pytest.mark.parametrize("prog", ["test_c_api", "test_c_api_extensions"])
def test_reference_c_test_programs(prog, tmp_path):
    """The reference's own C test programs (crates/matchy/tests/test_c_api*.c), compiled unmodified against the
    reference's header and linked with libmatchy_amd.so by `make -C oracle ref_c_tests` (needs /root/reference at build
    time; the binaries travel to the GPU box). They must pass as they do against the reference library."""
    exe = ROOT / "tests" / "ref_c_bin" / prog
    if not exe.exists():
        pytest.skip("tests/ref_c_bin not built (no /root/reference at build time)")
    r = subprocess.run([str(exe)], capture_output=True, timeout=300, cwd=tmp_path)
    out = r.stdout.decode("utf-8", "replace")
    assert r.returncode == 0, out[-2000:] + r.stderr.decode("utf-8", "replace")[-500:]
    assert "passed" in out


@pytest.mark.gpu
def test_query_subcommand(cli, tmp_path, oracle):
    """`matchy query` (bin/commands/query_cmd.rs:8-69): pretty JSON array, exit status 0 = found / 1 = not found."""
    import ipaddress
    import matchy_amd as M
    b = M.DatabaseBuilder(build_epoch=9)
    b.add_entry("192.0.2.0/24", {"net": "doc", "n": 24})
    b.add_entry("192.0.2.7", {"host": True})
    b.add_entry("2001:db8::/32", {"net": "doc6"})
    b.add_entry("evil.example.com", {"why": "literal", "tags": ["a", "b"], "nested": {"k": 1}})
    b.add_entry("*.example.com", {"why": "glob"})
    b.add_entry("*evil*", {"why": "glob2", "s": "quote\" and \\ and é"})
    b.add_entry("empty.example.org", {})
    dbp = tmp_path / "q.mxy"
    dbp.write_bytes(b.build())
    odb = oracle.Database(dbp.read_bytes())
    for q in ["192.0.2.7", "192.0.2.99", "198.51.100.1", "2001:db8::1", "2001:db9::1", "evil.example.com", "x.example.com", "devilish", "nothing.here",
              "empty.example.org", ""]:
        want = odb.lookup(q)
        if want["kind"] == "ip":
            net = ipaddress.ip_network(f"{q}/{want['prefix_len']}", strict=False)
            exp = [dict(want["data"], cidr=str(net), prefix_len=want["prefix_len"])]
            found = True
        elif want["kind"] == "pattern":
            exp = [d for d in want["data"] if d is not None]
            found = True
        else:
            exp, found = [], False
        r = _run([cli, "query", str(dbp), q])
        assert r.returncode == (0 if found else 1), (q, r.stderr)
        assert r.stdout.decode() == json.dumps(exp, indent=2, sort_keys=True, ensure_ascii=False) + "\n", q
        rq = _run([cli, "query", str(dbp), q, "-q"])
        assert rq.returncode == r.returncode and rq.stdout == b""
    assert _run([cli, "query", str(tmp_path / "missing.mxy"), "x"]).returncode == 1


_LINE_RANK = {"Domain": 0, "IPv4": 1, "Email": 2, "IPv6": 3, "Bitcoin": 5, "Ethereum": 6, "Monero": 7}


def _extract_expected(oracle, text, types=None, min_labels=2):
    """What extract_cmd.rs prints for `text`: LineScanner lines (trimmed of ASCII whitespace, empty ones skipped), per line the
    items of extract_from_line (lib.rs:1471-1521: domains, IPv4, e-mails, IPv6, hashes, Bitcoin, Ethereum, Monero)."""
    out = []
    ws = b" \t\n\r\x0c"
    for raw in text.split(b"\n"):
        line = raw.strip(ws)
        if not line:
            continue
        items = oracle.extract(line, min_labels=min_labels) if min_labels != 2 else oracle.extract(line)
        items = [it for it in items if types is None or it[0] in types or it[0] not in ("Domain", "IPv4", "Email", "IPv6")]
        items.sort(key=lambda it: (_LINE_RANK.get(it[0], 4), it[1]))
        out += [(t, line[s:e].decode()) for t, s, e, v in items]
    return out


@pytest.mark.gpu
def test_extract_subcommand(cli, tmp_path, oracle):
    """`matchy extract` (bin/commands/extract_cmd.rs): line order, formats, --types, --unique, margins trimmed like LineScanner."""
    from tools import synth
    cfg = synth.config("c1")
    text = synth.make_log(cfg, 0, 300)
    text += (b"\x0c 10.1.2.3 mail bob@example.org and 2001:db8::5 then evil.example.com 5d41402abc4b2a76b9719d911017c592 \x0c\r\n"
             b"\n   \n"
             b"x\x0c10.9.9.9 inner form feed is no boundary; caf\xc3\xa9.example.fr \"q\\uote.example.com\" 0x52908400098527886E0F7030069857D2E4169EE7\n"
             b"1BvBMSEYstWetqTFn5Au4m4GFg7xJaNVN2 last line without newline 8.8.8.8")
    p = tmp_path / "in.log"
    p.write_bytes(text)
    exp = _extract_expected(oracle, text)
    assert len({t for t, _ in exp}) >= 7
    r = _run([cli, "extract", str(p), "-s"])
    assert r.returncode == 0, r.stderr
    got = [json.loads(l) for l in r.stdout.decode().splitlines()]
    assert [(g["type"], g["value"]) for g in got] == [(t.lower(), v) for t, v in exp]
    n_lines = sum(1 for raw in text.split(b"\n") if raw.strip(b" \t\n\r\x0c"))
    assert f"[INFO] Lines processed: {n_lines:,}".encode() in r.stderr and f"[INFO] Patterns found: {len(exp):,}".encode() in r.stderr
    # small batches and stdin: same output
    r2 = _run([cli, "extract", "-", "--batch-bytes", "8192"], input=text)
    assert r2.returncode == 0 and r2.stdout == r.stdout
    # text and csv, unique
    rt = _run([cli, "extract", str(p), "--format", "text", "-u"])
    seen, uniq = set(), []
    for _, v in exp:
        if v not in seen:
            seen.add(v); uniq.append(v)
    assert rt.stdout.decode().splitlines() == uniq
    rc = _run([cli, "extract", str(p), "--format=csv", "--types", "ip"])
    lines = rc.stdout.decode().splitlines()
    exp_ip = _extract_expected(oracle, text, types=("IPv4", "IPv6"))
    assert lines[0] == "type,value" and lines[1:] == [f'{t.lower()},"{v}"' for t, v in exp_ip]
    assert any(t == "MD5" for t, _ in exp_ip) and not any(t == "Domain" for t, _ in exp_ip)
    # errors
    assert _run([cli, "extract", str(p), "--format", "xml"]).returncode == 1
    assert _run([cli, "extract", str(p), "--types", "bogus"]).returncode == 1
    assert _run([cli, "extract", str(tmp_path / "missing.log")]).returncode == 1


@pytest.mark.gpu
def test_match_follow(cli, tmp_path, oracle):
    """`matchy match -f` (match_processor/follow.rs): existing content first, then what is appended (also after a truncation),
    records stamped with the wall clock, statistics on SIGINT."""
    import signal
    import time
    from tools import synth
    cfg, rows = _c1_csv(tmp_path / "c1.csv")
    dbp = tmp_path / "c1.mxy"
    assert _run([cli, "build", str(tmp_path / "c1.csv"), "-o", str(dbp), "-f", "csv"]).returncode == 0
    odb = oracle.Database(dbp.read_bytes())
    part = [synth.make_log(cfg, a, 2000) for a in (0, 2000, 4000)]
    logp = tmp_path / "live.log"
    logp.write_bytes(part[0])
    assert _run([cli, "match", str(dbp), "-", "--follow"], input=b"").returncode == 1      # not with stdin
    outp = tmp_path / "out.ndjson"
    with open(outp, "wb") as out:
        pr = subprocess.Popen([cli, "match", str(dbp), str(logp), "--follow", "-s"], stdout=out, stderr=subprocess.PIPE)
        try:
            def wait_lines(n, timeout=60):
                t0 = time.time()
                while time.time() - t0 < timeout:
                    if len(outp.read_bytes().splitlines()) >= n:
                        return True
                    time.sleep(0.1)
                return False
            want = [odb.scan(p, source=str(logp))[1] for p in part]
            assert wait_lines(len(want[0]))
            with open(logp, "ab") as f:
                f.write(part[1])
            assert wait_lines(len(want[0]) + len(want[1]))
            logp.write_bytes(part[2])                      # truncated and rewritten: read again from the start
            assert wait_lines(len(want[0]) + len(want[1]) + len(want[2]))
        finally:
            pr.send_signal(signal.SIGINT)
            err = pr.communicate(timeout=60)[1]
    assert pr.returncode == 0, err
    got = [json.loads(l) for l in outp.read_bytes().decode().splitlines()]
    exp = [json.loads(l) for w in want for l in w]
    assert [dict(g, timestamp="") for g in got] == [dict(e, timestamp="") for e in exp]
    assert all(g["timestamp"] == "0.000" for g in got[:len(want[0])])
    assert all(float(g["timestamp"]) > 1.6e9 for g in got[len(want[0]):])
    assert b"Watching for new content" in err and b"Follow mode stopped" in err and b"[INFO] Lines processed: 6,000" in err


def test_inspect_and_validate(cli, tmp_path):
    """`matchy inspect` (inspect_cmd.rs; cli_tests.rs:163-205 ask for "Database:" and "Capabilities:") and `matchy validate`
    (exit status) — host-only subcommands."""
    (tmp_path / "p.txt").write_text("192.168.1.0/24\n*.evil.com\nbad.example.org\n")
    dbp = tmp_path / "t.mxy"
    r = _run([cli, "build", str(tmp_path / "p.txt"), "-o", str(dbp), "-d", "demo feed", "-t", "Threats"], env=dict(os.environ, MATCHY_BUILD_EPOCH="1700000000"))
    assert r.returncode == 0, r.stderr
    out = _run([cli, "inspect", str(dbp)]).stdout.decode()
    assert f"Database: {dbp}" in out and "Format:   Combined IP+String database" in out and "Capabilities:" in out
    assert "    Entries:       1" in out and "Literals:      ✓ (1 strings)" in out and "Globs:         ✓ (1 patterns)" in out
    assert "Database type:   Threats" in out and "    en: demo feed" in out and "Build time:      2023-11-14 22:13:20 UTC (1700000000)" in out
    assert "IP version:      IPv4" in out
    assert "Full metadata:" in _run([cli, "inspect", str(dbp), "-v"]).stdout.decode()
    js = json.loads(_run([cli, "inspect", str(dbp), "--json"]).stdout)
    assert js["file"] == str(dbp) and js["has_ip_data"] and js["has_glob_data"] and js["has_string_data"]
    assert (js["ip_count"], js["literal_count"], js["glob_count"]) == (1, 1, 1) and js["metadata"]["database_type"] == "Threats"
    assert _run([cli, "inspect", str(tmp_path / "missing.mxy")]).returncode == 1
    v = _run([cli, "validate", str(dbp)])
    assert v.returncode == 0 and b"VALIDATION PASSED" in v.stdout
    vj = json.loads(_run([cli, "validate", str(dbp), "--json", "--level", "standard"]).stdout)
    assert vj["is_valid"] is True and vj["errors"] == [] and vj["validation_level"] == "standard"
    bad = tmp_path / "bad.mxy"
    bad.write_bytes(dbp.read_bytes()[:200])
    vb = _run([cli, "validate", str(bad)])
    assert vb.returncode == 1 and b"VALIDATION FAILED" in vb.stdout
    assert json.loads(_run([cli, "validate", str(bad), "-j"]).stdout)["is_valid"] is False
    assert _run([cli, "validate", str(dbp), "--level", "paranoid"]).returncode == 1


@pytest.mark.gpu
def test_match_on_several_devices(cli, tmp_path, oracle):
    """`matchy match --devices 0,1` / `--devices all` on a box with at least two GPUs: line blocks go to one worker and
    scanner per device (database uploaded to each), output and counters equal the oracle's. Skipped on one-GPU boxes (the
    same code path runs there with one device listed several times: test_match_config1_end_to_end)."""
    import matchy_amd as M
    from tools import synth
    if M.lib().matchy_amd_device_count() < 2:
        pytest.skip("needs at least two GPUs")
    cfg, rows = _c1_csv(tmp_path / "c1.csv")
    dbp = tmp_path / "c1.mxy"
    assert _run([cli, "build", str(tmp_path / "c1.csv"), "-o", str(dbp), "-f", "csv"]).returncode == 0
    log = synth.make_log(cfg, 0, 60000)
    logp = tmp_path / "access.log"
    logp.write_bytes(log)
    want_hits, want_lines, st = oracle.Database(dbp.read_bytes()).scan(log, source=str(logp))
    for devs in ("0,1", "all", "1"):
        r = _run([cli, "match", str(dbp), str(logp), "--devices", devs, "--batch-bytes", "400000", "-s"])
        assert r.returncode == 0, r.stderr
        assert r.stdout.decode().splitlines() == want_lines
        assert f"[INFO] Candidates tested: {st.candidates:,}".encode() in r.stderr
