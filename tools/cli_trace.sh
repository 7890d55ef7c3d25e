#!/bin/bash
# stage timings of `matchy match` on the GPU box: tools/cli_trace.sh [lines] [reps]
L=${1:-10000000}; R=${2:-3}
python - <<PY
import sys
sys.path.insert(0, ".")
from tools import synth
cfg = synth.config("c2")
open("/tmp/c2.mxy", "wb").write(synth.build_db(cfg))
with open("/tmp/c2.log", "wb") as f:
    for a in range(0, $L, 1_000_000):
        f.write(synth.make_log(cfg, a, min(1_000_000, $L - a)))
PY
files=""; for i in $(seq $R); do files="$files /tmp/c2.log"; done
for devs in 0 0,0; do
  echo "== devices $devs"
  MATCHY_AMD_TRACE=1 matchy_amd/bin/matchy match /tmp/c2.mxy $files --devices $devs --batch-bytes $((256<<20)) --format summary -s 2>&1 >/dev/null | grep -E "batch|scan_host|fetch:|Throughput|Processing time" | head -${3:-24}
done
head -c 1000000 /tmp/c2.log > /tmp/small.log
echo "== fixed cost (1 MB input)"
for i in 1 2 3; do /usr/bin/time -f "%e s wall" matchy_amd/bin/matchy match /tmp/c2.mxy /tmp/small.log --format summary -s 2>&1 >/dev/null | grep -E "wall|Processing time" ; done
echo "== 30 x 1.8 GB"
files=""; for i in $(seq 30); do files="$files /tmp/c2.log"; done
for devs in 0 0,0 0,0,0; do /usr/bin/time -f "%e s wall" matchy_amd/bin/matchy match /tmp/c2.mxy $files --devices $devs --batch-bytes $((256<<20)) --format summary -s 2>&1 >/dev/null | grep -E "wall|Throughput|Processing time"; done
