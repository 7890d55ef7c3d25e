#!/bin/bash
# Build a kernel variant and keep it for an A/B run: tools/abv.sh NAME [extra compiler flags, e.g. -DMXY_V4_SECOND_DOT]
# -> matchy_amd/lib_ab/NAME.so (git-ignored, travels to the GPU box; select with MATCHY_AMD_LIB, see tools/ab.sh). Only k_anchor.hip
# is recompiled with the flags; the other objects of the last regular build are linked as they are.
set -e
cd "$(dirname "$0")/.."
NAME=$1; shift
mkdir -p matchy_amd/lib_ab /tmp/abv
SRC=${ABV_SRC:-k_anchor.hip}
OBJ=/tmp/abv/$NAME.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -Wall -Wno-unused-function -Wno-unused-result -DNDEBUG "$@" -x hip -c matchy_amd/csrc/$SRC -o $OBJ
OBJS=""
for o in matchy_amd/lib/*.o; do
  if [ "$(basename $o)" = "${SRC%.*}.o" ]; then OBJS="$OBJS $OBJ"; else OBJS="$OBJS $o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o matchy_amd/lib_ab/$NAME.so $OBJS -ldl -lpthread
echo saved matchy_amd/lib_ab/$NAME.so
