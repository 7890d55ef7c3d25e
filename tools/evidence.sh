#!/bin/bash
# Regenerates the per-round evidence under gpurun_out/ev_<tag>/ on a GPU box (repo root).
#   A = headline: rocprofv3 trace + PMC passes (tools/prof.sh), the default bench line, kernel timeline of one step
#   B = the other single-GPU configurations (one 10 M-line batch per step), strong scaling on one GPU
#   C = command line end to end, host-to-device strategies, hostile inputs, fuzz campaign
# Usage: tools/evidence.sh <tag> A|B|C
TAG=$1; PART=$2
export TMPDIR=/tmp
O=gpurun_out/ev_$TAG
mkdir -p $O
if [ "$PART" = A ]; then
  bash tools/prof.sh $TAG > $O/prof.log 2>&1 || exit 1
  timeout -k 10 500 python bench.py > $O/bench_c2.json 2> $O/bench_c2.err || exit 1
  bash tools/tl.sh $TAG > $O/step_timeline.txt 2>&1 || exit 1
elif [ "$PART" = B ]; then
  for c in c3 c3b c4 c5; do
    timeout -k 10 400 python bench.py --config $c --lines 10000000 --steps 10 --cpu-lines 1000000 --no-e2e > $O/bench_${c}_10Mlines.json 2> $O/bench_$c.err || exit 1
  done
  timeout -k 10 600 python bench.py --scaling strong --lines 20000000 --steps 2 --warmup 1 > $O/strong_scaling_1gpu_20Mlines.json 2> $O/strong.err || exit 1
else
  timeout -k 10 300 python tools/cli_fixed.py 30 > $O/cli_end_to_end.txt 2>&1 || exit 1
  timeout -k 10 300 python tools/cli_e2e.py 10000000 6 >> $O/cli_end_to_end.txt 2>&1 || exit 1
  hipcc -O2 -o /tmp/h2d_rate2 tools/ubench/h2d_rate2.cpp -lpthread > /dev/null 2>&1 && timeout -k 10 200 /tmp/h2d_rate2 /tmp/c2.log > $O/h2d_rates.txt 2>&1
  timeout -k 10 300 python tools/hostile_inputs.py > $O/hostile_inputs.txt 2>&1 || exit 1
  timeout -k 10 400 python tools/fuzz_campaign.py 240 3000 > $O/fuzz_campaign.txt 2>&1 || exit 1
fi
echo done $PART
