#!/bin/bash
# One bench run per line of knobs: tools/sweep_tail_knobs.sh "<name> <shape> ENV=VAL ..." ...   (GPU box, repo root)
O=gpurun_out/gridsweep; mkdir -p $O
for spec in "$@"; do
  set -- $spec
  name=$1; sh=$2; shift 2
  timeout -k 10 200 env X=1 "$@" python bench.py --log-shape $sh --no-cpu --steps 20 --no-e2e --no-scatter-gather > $O/$name.json 2> $O/$name.err || { echo FAILED $name; exit 1; }
  python3 - $name $O/$name.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
k = list(d["kernel_ms"].values())
print(f"{sys.argv[1]:28s} step {d['ms_per_step']:7.3f} ms {d['value']:7.1f} GB/s k_anchor {k[0]:6.3f} tail {k[1]:6.3f} pipelined {d['pipelined']['value']:7.1f}", flush=True)
PY
done
