#!/bin/bash
# Usage (on the GPU box, from the repo root): tools/prof.sh <tag> [bench args...]
# Runs bench.py under rocprofv3 three times (kernel trace + two PMC passes; never combined, see the task notes)
# and leaves CSVs plus a summary under gpurun_out/prof_<tag>/.
set -e
TAG=$1; shift
export TMPDIR=/tmp
R=$PWD
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT/trace $OUT/pmc1 $OUT/pmc2 $OUT/pmc3
ARGS="--steps 3 --warmup 1 --no-cpu --no-e2e --no-scatter-gather --pipelined 0 $@"
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/trace.log 2>&1
timeout -k 10 280 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY --output-format csv -d $OUT/pmc1 -- python3 bench.py $ARGS > $OUT/pmc1.log 2>&1
timeout -k 10 280 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc2 -- python3 bench.py $ARGS > $OUT/pmc2.log 2>&1
timeout -k 10 280 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc3 -- python3 bench.py $ARGS > $OUT/pmc3.log 2>&1
timeout -k 10 280 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc4 -- python3 bench.py $ARGS > $OUT/pmc4.log 2>&1
python3 tools/prof_summary.py $OUT > $OUT/summary.txt
cat $OUT/summary.txt
