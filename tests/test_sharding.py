"""N>1 path of bench.py on CPU: world-size-2 gloo run of the line-block sharding.

Each rank generates ITS OWN block of the synthetic log (counter-based generator), scans it with the CPU oracle (the
checker — the product path needs a GPU), and the ranks reduce their scalars exactly as bench.py does
(matchy_amd.sharding.aggregate). Rank 0 then checks the sharded totals against one scan of the whole log:
N4 (no candidate crosses a line feed) makes the two equal.
"""
import json
import os
import socket
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent

WORKER = r"""
import json, os, sys, time
sys.path.insert(0, os.environ["MXY_ROOT"])
import torch, torch.distributed as dist
from matchy_amd import sharding
from tools import synth
from oracle import oracle

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group(backend="gloo")
L = int(os.environ["MXY_LINES"])
cfg = synth.config("c1")
blob = open(os.environ["MXY_DB"], "rb").read()
blk = sharding.block_for_rank(rank, world, L)
log = synth.make_log(cfg, blk.first_line, blk.n_lines)
db = oracle.Database(blob)
sharding.barrier(dist, world)
t0 = time.perf_counter()
hits, _, st = db.scan(log, threads=1, cache=0, want_json=False)
sharding.barrier(dist, world)
el = time.perf_counter() - t0
agg = sharding.aggregate(dist, world, torch.device("cpu"), el, len(log), st.lines, len(hits), st.candidates)
# every rank must hold the same reduced values
gathered = [None] * world
dist.all_gather_object(gathered, agg)
assert all(g == gathered[0] for g in gathered), gathered
if rank == 0:
    whole = synth.make_log(cfg, 0, L * world)
    ghits, _, gst = db.scan(whole, threads=1, cache=0, want_json=False)
    print(json.dumps({"agg": agg, "whole": {"bytes": len(whole), "lines": gst.lines, "hits": len(ghits), "candidates": gst.candidates},
                      "own_elapsed": el}))
dist.destroy_process_group()
"""


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_block_for_rank_and_split():
    from matchy_amd import sharding
    assert sharding.block_for_rank(0, 8, 1000).first_line == 0
    assert sharding.block_for_rank(7, 8, 1000).first_line == 7000
    with pytest.raises(ValueError):
        sharding.block_for_rank(8, 8, 10)
    data = b"aa\nbbbb\nc\n\ndddddd\nee"
    for parts in (1, 2, 3, 5, 9):
        rs = sharding.split_at_newlines(data, parts)
        assert len(rs) == parts and rs[0][0] == 0 and rs[-1][1] == len(data)
        for (a, b), (c, d) in zip(rs, rs[1:]):
            assert b == c and a <= b
        for a, b in rs[:-1]:
            assert b == a or b == len(data) or data[b - 1:b] == b"\n"
    assert sharding.split_at_newlines(b"", 3) == [(0, 0)] * 3
    assert sharding.split_at_newlines(b"no newline at all", 2) == [(0, 17), (17, 17)]


def test_strong_blocks_cover_the_job_once():
    from matchy_amd import sharding
    for total, world in ((100, 8), (7, 3), (5, 8), (100_000_000, 8), (1, 1)):
        blocks = [sharding.strong_block_for_rank(r, world, total) for r in range(world)]
        assert blocks[0].first_line == 0
        for a, b in zip(blocks, blocks[1:]):
            assert a.first_line + a.n_lines == b.first_line
        assert blocks[-1].first_line + blocks[-1].n_lines == total
        assert max(b.n_lines for b in blocks) - min(b.n_lines for b in blocks) <= 1
    with pytest.raises(ValueError):
        sharding.strong_block_for_rank(3, 3, 10)


def test_aggregate_single_rank():
    from matchy_amd import sharding
    a = sharding.aggregate(None, 1, None, 0.5, 10, 2, 1, 3)
    assert a == {"elapsed_s": 0.5, "bytes": 10, "lines": 2, "hits": 1, "candidates": 3}


def test_world_size_2_gloo(oracle, tmp_path):
    from tools import synth
    cfg = synth.config("c1")
    # database built through the oracle-independent host builder is GPU-free (pure C++), so it works here
    dbp = tmp_path / "c1.mxy"
    dbp.write_bytes(synth.build_db(cfg))
    script = tmp_path / "worker.py"
    script.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   MXY_ROOT=str(ROOT), MXY_LINES="1500", MXY_DB=str(dbp), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(script)], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    for p in procs:
        try:
            o, e = p.communicate(timeout=240)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append((p.returncode, o, e))
    for rc, o, e in outs:
        assert rc == 0, e[-2000:]
    rep = json.loads(outs[0][1].strip().splitlines()[-1])
    agg, whole = rep["agg"], rep["whole"]
    assert agg["bytes"] == whole["bytes"]
    assert agg["lines"] == whole["lines"] == 3000
    assert agg["hits"] == whole["hits"] and agg["hits"] > 0
    assert agg["candidates"] == whole["candidates"]
    assert agg["elapsed_s"] >= rep["own_elapsed"] - 1e-9  # MAX over ranks


def _bench(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    e.update(env)
    return subprocess.run([sys.executable, str(ROOT / "bench.py"), *args], env=e, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)


def test_bench_gpus_flag_is_what_runs():
    """`python bench.py --gpus 2` without a launcher starts two ranks itself (child torch.distributed.run, gloo here) and
    reports n_gpus = 2; a launcher whose world differs from --gpus is refused; without enough devices it fails loudly."""
    r = _bench(["--gpus", "2", "--steps", "3", "--lines", "1000"], BENCH_DRYRUN="1", OMP_NUM_THREADS="1")
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["lines"] == 2000 and line["dry_run"] is True
    r = _bench(["--gpus", "4"], WORLD_SIZE="2", RANK="0", BENCH_DRYRUN="1")
    assert r.returncode != 0 and "--gpus 4" in r.stderr
    # no GPU in the CPU container: the real path must refuse instead of printing a line for fewer devices than asked for
    import torch
    if torch.cuda.device_count() < 2:
        r = _bench(["--gpus", "2", "--steps", "1"], OMP_NUM_THREADS="1")
        assert r.returncode != 0
        assert "device(s) are visible" in r.stderr
        assert not [l for l in r.stdout.splitlines() if l.startswith("{")]
