export TMPDIR=/tmp
OUT=$PWD/gpurun_out/pmc_win1
rm -rf $OUT; mkdir -p $OUT
timeout -k 10 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_FLAT SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $OUT -- python3 bench.py --steps 2 --warmup 1 --no-cpu --no-e2e --no-scatter-gather --pipelined 0 --log-shape $1 > $OUT.log 2>&1
python3 - $OUT <<'PY'
import csv, glob, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"][:40]
        if "k_validate" in k or "k_lookup" in k:
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k,v in agg.items():
    print(k, {c: round(sum(x)/len(x)/1e6,2) for c,x in sorted(v.items())})
PY
