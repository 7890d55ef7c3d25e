#!/bin/bash
# Per-kernel instruction and wave-cycle counters of one bench step: tools/pmc_kernel.sh <shape> [tag] [more bench args, e.g. --config c3b]   (GPU box, repo root)
# One rocprofv3 --pmc pass (no trace domains beside it). Millions per launch, averaged over the launches of the run.
export TMPDIR=/tmp
SH=$1; TAG=${2:-$1}; shift; shift
OUT=$PWD/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-e2e --no-scatter-gather --pipelined 0 --log-shape $SH "$@" > $OUT.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"]
        if "mxy::" not in k or "k_ip_" in k: continue
        k = re.sub(r"\(.*", "", k).replace("void ", "").replace("mxy::", "")
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        agg[k]["_wgs"].append(float(r["Grid_Size"]) / float(r["Workgroup_Size"])); agg[k]["_lds"].append(float(r["LDS_Block_Size"])); agg[k]["_vgpr"].append(float(r["VGPR_Count"]))
for k, v in agg.items():
    print(f"{k:24s}", {c.replace("SQ_", ""): round(sum(x) / len(x) / (1 if c[0] == '_' else 1e6), 2) for c, x in sorted(v.items())})
PY
