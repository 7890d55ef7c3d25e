#!/bin/bash
# Headline-batch sweep on the GPU box (repo root): one short bench.py run per "ENV=... [bench args]" line on stdin, one result line each.
# Usage: tools/sweep.sh <tag> < variants.txt     (a line may start with NAME: )
TAG=$1
O=gpurun_out/sweep_$TAG
mkdir -p $O
i=0
while IFS= read -r line; do
  [ -z "$line" ] && continue
  i=$((i+1))
  name=${line%%:*}; spec=${line#*:}
  # spec = "VAR=value ... [:: bench args]"
  envs=${spec%%::*}; args=""; [[ "$spec" == *::* ]] && args="${spec#*::}"
  ( eval "env $envs timeout -k 10 280 python bench.py --no-cpu --no-e2e --no-scatter-gather --pipelined 0 --steps 30 --warmup 3 $args" ) > $O/$i.json 2> $O/$i.err
  python3 - "$name" $O/$i.json <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[2]) if l.startswith("{")][-1])
    print(f"{sys.argv[1]:28s} {d['ms_per_step']:.4f} ms  {d['value']:.1f} GB/s  slices={d.get('slices')}  kernel_ms={list(d['kernel_ms'].values())}  dev={d['device_results']['ms_per_step']}", flush=True)
except Exception as e:
    print(sys.argv[1], "FAILED", e, flush=True)
PY
done
