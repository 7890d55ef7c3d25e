#!/bin/bash
# kernel timeline of one bench step on the GPU box: tools/tl.sh <tag> [bench args]
TAG=$1; shift
export TMPDIR=/tmp
O=$PWD/gpurun_out/tl_$TAG
rm -rf $O; mkdir -p $O
timeout -k 10 280 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --steps 4 --warmup 1 --no-cpu --no-e2e --no-scatter-gather --pipelined 0 "$@" > $O/bench.log 2>&1
python3 tools/timeline.py $O > $O/timeline.txt
cat $O/timeline.txt
