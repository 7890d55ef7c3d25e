#!/bin/bash
# A/B of library builds under rocprofv3 --pmc: tools/pmc_ab.sh NAME... (libraries saved with tools/ab.sh save NAME) — prints k_anchor's
# instruction counters for each
export TMPDIR=/tmp
for v in "$@"; do
  OUT=$PWD/gpurun_out/pmc_ab_$v
  rm -rf $OUT; mkdir -p $OUT
  MATCHY_AMD_LIB=$PWD/matchy_amd/lib_ab/$v.so MATCHY_AMD_PSL=$PWD/matchy_amd/data/psl.bin timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-e2e --no-scatter-gather --pipelined 0 > $OUT.log 2>&1
  python3 - $OUT $v <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_anchor" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print(sys.argv[2], {k: round(sum(v) / len(v) / 1e6, 1) for k, v in sorted(agg.items())})
PY
done
