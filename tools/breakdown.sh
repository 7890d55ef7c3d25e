#!/bin/bash
# Instruction breakdown of k_anchor with a throw-away instrumented build (MATCHY_AMD_CFLAGS=-DMXY_ANCHOR_DEBUG python -m matchy_amd.build
# --force; tools/ab.sh save dbg; rebuild): for every MATCHY_AMD_DEBUG value given, one rocprofv3 --pmc pass of bench.py; prints VALU / SALU /
# LDS wave-instructions and the kernel time. Usage (GPU box, repo root): tools/breakdown.sh <tag> 0 1 2 4 8 ...
TAG=$1; shift
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/bd_$TAG
mkdir -p $OUT
export MATCHY_AMD_LIB=$PWD/matchy_amd/lib_ab/dbg.so MATCHY_AMD_PSL=$PWD/matchy_amd/data/psl.bin
for D in "$@"; do
  MATCHY_AMD_DEBUG=$D timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_BUSY_CYCLES --output-format csv -d $OUT/d$D -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-e2e --no-scatter-gather --pipelined 0 > $OUT/d$D.log 2>&1
  python3 - $OUT/d$D $D <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_anchor" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("debug", sys.argv[2], {k: round(sum(v) / len(v) / 1e6, 1) for k, v in sorted(agg.items())})
PY
  grep -o '"k_anchor": [0-9.]*' $OUT/d$D.log | head -1
done
