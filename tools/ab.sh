#!/bin/bash
# A/B of kernel variants: tools/ab.sh save NAME   copies the built library to matchy_amd/lib_ab/NAME.so (git-ignored, travels to the
# GPU box); a sweep line then selects it with MATCHY_AMD_LIB=matchy_amd/lib_ab/NAME.so MATCHY_AMD_PSL=matchy_amd/data/psl.bin
set -e
cd "$(dirname "$0")/.."
case "$1" in
  save) mkdir -p matchy_amd/lib_ab; cp matchy_amd/lib/libmatchy_amd.so matchy_amd/lib_ab/$2.so; echo saved matchy_amd/lib_ab/$2.so ;;
  *) echo "usage: tools/ab.sh save NAME"; exit 1 ;;
esac
