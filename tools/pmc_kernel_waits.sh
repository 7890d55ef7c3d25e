export TMPDIR=/tmp
SH=$1; KN=$2
for set in "SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_SALU SQ_INSTS_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_ANY"; do
OUT=$PWD/gpurun_out/pmc3_$SH; rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --pmc $set --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-e2e --no-scatter-gather --pipelined 0 --log-shape $SH > $OUT.log 2>&1
python3 - $OUT $KN <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(list)
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print({k.replace("SQ_",""): round(sum(v)/len(v)/1e6,1) for k,v in sorted(agg.items())})
PY
done
