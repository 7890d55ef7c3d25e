#!/bin/bash
# Grid sizes of the k_validate variants on the dense log shapes: tools/sweep_misc_grid.sh "<shape> <vmode> <grid> ..." ...
# (MATCHY_AMD_MISC_GRID_V<vmode>, csrc/validate_kernels.hip launch_validate_misc; vmode 4 = long tokens, 1 = IPv6 / e-mail anchors,
# 2 = domains k_validate_dom left undecided). GPU box, repo root.
O=gpurun_out/gridsweep; mkdir -p $O
for spec in "$@"; do
  set -- $spec
  sh=$1; vm=$2; shift 2
  for g in "$@"; do
    timeout -k 10 200 env MATCHY_AMD_MISC_GRID_V$vm=$g python bench.py --log-shape $sh --no-cpu --steps 20 --no-e2e --no-scatter-gather > $O/$sh.$vm.$g.json 2> $O/$sh.$vm.$g.err || { echo FAILED $sh $vm $g; exit 1; }
    python3 - $sh $vm $g $O/$sh.$vm.$g.json <<'PY'
import json, sys
d = json.loads([l for l in open(sys.argv[4]) if l.startswith("{")][-1])
k = list(d["kernel_ms"].values())
print(f"{sys.argv[1]:12s} V{sys.argv[2]} grid {sys.argv[3]:5s} step {d['ms_per_step']:7.3f} ms {d['value']:7.1f} GB/s k_anchor {k[0]:6.3f} tail {k[1]:6.3f} pipelined {d['pipelined']['value']:7.1f}", flush=True)
PY
  done
done
