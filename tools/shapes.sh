#!/bin/bash
# Throughput on other log shapes (bench.py --log-shape, same-run parity against the oracle for each): tools/shapes.sh <tag> [extra bench args]
TAG=$1; shift
O=gpurun_out/shapes_$TAG
mkdir -p $O
for sh in nginx jsonl-app ip-dense url-heavy hash-dense skewed-halves; do
  timeout -k 10 300 python bench.py --log-shape $sh --cpu-lines 500000 --steps 20 --no-e2e --no-scatter-gather "$@" > $O/$sh.json 2> $O/$sh.err || echo FAILED $sh
  python3 - $sh $O/$sh.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
k = list(d["kernel_ms"].values())
print(f"{sys.argv[1]:14s} lines {d['config']['lines_per_gpu']:9d}  bytes {d['config']['bytes_per_gpu']:10d}  step {d['ms_per_step']:8.3f} ms  {d['value']:7.1f} GB/s  k_anchor {k[0]:7.3f}  tail {k[1]:7.3f}  "
      f"cands {d['candidates_per_step']:9d}  hits {d['hits_per_step']:8d}  parity {d['parity_vs_oracle']}  pipelined {d['pipelined']['value']:7.1f} GB/s  cpu16 {d['cpu_baseline']['value']:.2f} GB/s", flush=True)
PY
done
