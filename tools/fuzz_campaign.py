"""Differential fuzz campaign on the GPU box: many more seeds than the committed tests run (same generators + a mutation
fuzzer over the synthetic logs), GPU path against the oracle. Usage: python tools/fuzz_campaign.py [seconds] [first_seed]"""
import importlib.util
import random
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import matchy_amd as M  # noqa: E402
from oracle import oracle  # noqa: E402
from tools import synth  # noqa: E402

spec = importlib.util.spec_from_file_location("tgp", ROOT / "tests" / "test_gpu_parity.py")
tgp = importlib.util.module_from_spec(spec)
spec.loader.exec_module(tgp)

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
oracle.build()
oracle.lib()
ex = M.Extractor()
cfg = synth.config("c4/50")
blob = synth.build_db(cfg)
db = M.Database(blob)
sc = M.Scanner(db)
odb = oracle.Database(blob)
base_log = synth.make_log(cfg, 0, 3000)
lines = base_log.split(b"\n")


def check_extract(buf, what):
    got, want = tgp.norm(ex.extract_from_chunk(buf)), tgp.norm(oracle.extract(buf))
    if got != want:
        Path("gpurun_out").mkdir(exist_ok=True)
        Path("gpurun_out/fuzz_fail.bin").write_bytes(buf)
        diff = sorted(set(map(tuple, got)) ^ set(map(tuple, want)))[:10]
        print(f"MISMATCH extract [{what}]: {len(got)} vs {len(want)}; first differences {diff}", flush=True)
        return False
    return True


import ctypes  # noqa: E402
hip = ctypes.CDLL("libamdhip64.so")


def check_scan(buf, what):
    res = sc.scan(buf)
    got = res.hits()
    gs = (res.lines, res.candidates)
    res.close()
    if buf:   # the device-resident entry (independent parts of the scan on three streams) must agree with the host-buffer entry
        dptr = ctypes.c_void_p()
        assert hip.hipMalloc(ctypes.byref(dptr), ctypes.c_size_t(len(buf) + 64)) == 0
        assert hip.hipMemcpy(dptr, bytes(buf), ctypes.c_size_t(len(buf)), 1) == 0
        rd = sc.scan_device(dptr.value, len(buf), fetch_mode=3)
        same = rd.hits() == got and (rd.lines, rd.candidates) == gs
        rd.close()
        hip.hipFree(dptr)
        if not same:
            Path("gpurun_out").mkdir(exist_ok=True)
            Path("gpurun_out/fuzz_fail.bin").write_bytes(buf)
            print(f"MISMATCH device-resident vs host-buffer entry [{what}]", flush=True)
            return False
    want, _, st = odb.scan(buf, want_json=False)
    if got != want or gs != (st.lines, st.candidates):
        Path("gpurun_out").mkdir(exist_ok=True)
        Path("gpurun_out/fuzz_fail.bin").write_bytes(buf)
        print(f"MISMATCH scan [{what}]: hits {len(got)} vs {len(want)}, stats {gs} vs {(st.lines, st.candidates)}", flush=True)
        return False
    return True


t0 = time.time()
seed = seed0
n = 0
ok = True
last = t0
while ok and time.time() - t0 < budget:
    rng = random.Random(seed)
    kind = seed % 6
    if kind == 0:
        ok = check_extract(tgp._long_domain_runs(seed, 150), f"long domains seed {seed}")
    elif kind == 1:
        ok = check_extract(tgp._long_emails(seed, 150), f"long emails seed {seed}")
    elif kind == 2:
        for _ in range(40):
            alpha = rng.choice(tgp.ALPHABETS)
            m = rng.choice([1, 5, 17, 63, 64, 65, 127, 128, 129, 200, 1000, 1023, 1024, 1025, 5000, 20000])
            ok = ok and check_extract(bytes(rng.choice(alpha) for _ in range(m)), f"alphabet seed {seed}")
    elif kind == 3:
        buf = tgp._mutated_log(seed, lines)
        ok = check_extract(buf, f"mutated log seed {seed}") and check_scan(buf, f"mutated log seed {seed}")
    elif kind == 5:
        entries, log, _ = tgp._ci_case(seed)
        log = log + tgp._mutated_log(seed, log.split(b"\n"), 100)
        b = M.DatabaseBuilder(build_epoch=7, case_insensitive=True)
        for k, v in entries:
            b.add_entry(k, v)
        gh, gl, gs, wh, wl, ws = tgp._scan_both(M, oracle, b.build(), log)
        b.close()
        if not (gs == ws and gh == wh):
            print(f"MISMATCH case-insensitive seed {seed}: {len(gh)} vs {len(wh)}", flush=True)
            ok = False
    else:
        pats, log = tgp._glob_fuzz_case(seed)
        b = M.DatabaseBuilder(build_epoch=6)
        for pt, i in pats.items():
            b.add_entry(pt, {"g": i})
        gh, gl, gs, wh, wl, ws = tgp._scan_both(M, oracle, b.build(), log)
        b.close()
        if not (gs == ws and gh == wh):
            print(f"MISMATCH glob fuzz seed {seed}: {len(gh)} vs {len(wh)}", flush=True)
            ok = False
    n += 1
    seed += 1
    if time.time() - last > 30:
        print(f"... {n} cases, {time.time() - t0:.0f} s", flush=True)
        last = time.time()
print(f"{'OK' if ok else 'FAILED'}: {n} cases (seeds {seed0}..{seed - 1}) in {time.time() - t0:.0f} s", flush=True)
sys.exit(0 if ok else 1)
