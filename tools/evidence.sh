#!/bin/bash
# Regenerates the per-round evidence under gpurun_out/ev_<tag>/ on a GPU box (repo root): part A = headline profile + bench +
# the other single-GPU configurations, part B = the 10 M-indicator database, strong scaling, hostile inputs, CLI end to end.
# Usage: tools/evidence.sh <tag> A|B
TAG=$1; PART=$2
export TMPDIR=/tmp
O=gpurun_out/ev_$TAG
mkdir -p $O
if [ "$PART" = A ]; then
  bash tools/prof.sh $TAG > $O/prof.log 2>&1 || exit 1
  timeout -k 10 400 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err || exit 1
  timeout -k 10 300 python bench.py --pipelined 3 --no-cpu --no-e2e > $O/bench_c2_pipelined.json 2> $O/bench_c2_pipelined.err || exit 1
  for c in c3 c3b c4; do
    timeout -k 10 400 python bench.py --config $c --lines 10000000 --steps 10 --cpu-lines 1000000 --no-e2e > $O/bench_${c}_10Mlines.json 2> $O/bench_$c.err || exit 1
  done
elif [ "$PART" = C4 ]; then
  timeout -k 10 400 python bench.py --config c4 --lines 10000000 --steps 10 --cpu-lines 1000000 --no-e2e > $O/bench_c4_10Mlines.json 2> $O/bench_c4.err || exit 1
else
  timeout -k 10 900 python bench.py --config c5 --lines 10000000 --steps 10 --cpu-lines 1000000 --no-e2e --pipelined 3 > $O/bench_c5_10Mlines.json 2> $O/bench_c5.err || exit 1
  timeout -k 10 600 python bench.py --scaling strong --lines 20000000 --steps 2 --warmup 1 > $O/strong_scaling_1gpu_20Mlines.json 2> $O/strong.err || exit 1
  timeout -k 10 300 python tools/hostile_inputs.py > $O/hostile_inputs.txt 2>&1 || exit 1
  timeout -k 10 300 python tools/cli_e2e.py 10000000 6 > $O/cli_end_to_end.txt 2>&1 || exit 1
fi
echo done $PART
