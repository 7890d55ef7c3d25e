#!/bin/bash
# variants of the command-line pipeline, 30 x 1.8 GB, --format summary: tools/cli_var.sh
[ -f /tmp/c2.log ] || python tools/cli_fixed.py 1 > /dev/null
files=""; for i in $(seq ${REPS:-30}); do files="$files /tmp/c2.log"; done
run() { # name, env..., -- args
  name=$1; shift
  s=$(date +%s%N)
  out=$(env "$@" MATCHY_AMD_TRACE=1 matchy_amd/bin/matchy match /tmp/c2.mxy $files --format summary -s $ARGS 2>&1 >/dev/null | grep -E "Throughput|database open|all batches|cleaned" | tr "\n" " ")
  e=$(date +%s%N)
  echo "$name: wall $(( (e-s)/1000000 )) ms; $out"
}
ARGS="--devices 0,0 --batch-bytes $((256<<20))"; run "2 workers 256M" A=1
ARGS="--devices 0,0,0,0 --batch-bytes $((256<<20))"; run "4 workers 256M" A=1
ARGS="--devices 0,0,0,0,0,0 --batch-bytes $((256<<20))"; run "6 workers 256M" A=1
ARGS="--devices 0,0,0,0,0,0,0,0 --batch-bytes $((128<<20))"; run "8 workers 128M" A=1
ARGS="--devices 0,0,0,0,0,0 --batch-bytes $((128<<20))"; run "6 workers 128M" A=1
ARGS="--devices 0,0,0 --batch-bytes $((256<<20))"; run "3 workers 256M" A=1
