#!/bin/bash
# A/B tooling: build a variant of libsitrk.so next to the product's own (never loaded unless SITRK_LIB_PATH points at it).
#   tools/build_variant.sh <name> [-DSITRK_... extra hipcc flags]      -> build_ab/libsitrk_<name>.so
# The rocPRIM sort object is shared with the product build (make -C sitrack_amd/csrc first).
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SRC=${SRC:-$ROOT/sitrack_amd/csrc}
mkdir -p $ROOT/build_ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fno-fast-math -fvisibility=hidden -Wall \
    -Wno-unused-function "$@" -c -o $ROOT/build_ab/sitrk_$NAME.o $SRC/sitrk.hip
[ -f $SRC/sitrk_sort.o ] || make -C $SRC sitrk_sort.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -o $ROOT/build_ab/libsitrk_$NAME.so $ROOT/build_ab/sitrk_$NAME.o $SRC/sitrk_sort.o
echo built $ROOT/build_ab/libsitrk_$NAME.so
