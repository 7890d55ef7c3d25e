"""Writes the golden match sets of the BASELINE configs (SURVEY §8c "Consequence"): C1 in full (1 K indicators, 10 K lines),
the first 1 000 lines of C2, C3, C3b, C4 (full-size databases) and C5 (indicators scaled 1/100: the full 10 M-entry database
takes 20 s to build). Producer: the CPU oracle (oracle/), i.e. our restatement of the reference CPU path — these files pin
the oracle against itself over time and give the GPU tests a fixed target; they are not outputs of the reference binary
(Rust cannot be built here). Inputs are regenerated from tools/synthgen.cpp (counter-based, seed in tools/synth.py).

Usage: python tests/golden/make_config_golden.py
"""
import sys
from pathlib import Path

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE.parent.parent))
from oracle import oracle  # noqa: E402
from tools import synth  # noqa: E402

CASES = [("c1", 10000), ("c2", 1000), ("c3", 1000), ("c3b", 1000), ("c4", 1000), ("c5/100", 1000)]


def golden_path(cfgname):
    return HERE / f"config_{cfgname.replace('/', '_')}.ndjson"


def produce(cfgname, lines):
    cfg = synth.config(cfgname)
    blob = synth.build_db(cfg)
    log = synth.make_log(cfg, 0, lines)
    hits, ndjson, st = oracle.Database(blob).scan(log, source="access.log")
    header = f'{{"config":"{cfgname}","lines":{st.lines},"candidates":{st.candidates},"matches":{len(ndjson)}}}'
    return header, ndjson


if __name__ == "__main__":
    oracle.build()
    for name, lines in CASES:
        header, nd = produce(name, lines)
        golden_path(name).write_text("\n".join([header] + nd) + "\n")
        print(golden_path(name).name, header)
