#!/bin/bash
# Build a variant of the hot gather kernels only: ucg_pair_hot.hip compiled with extra flags (e.g. -DUCG_VARIANT=3) and linked
# with the objects of the regular build into _ab/libs/libucg_<name>.so; run with UCG_HIP_LIBRARY=_ab/libs/libucg_<name>.so.
# usage: tools/build_variant.sh <name> [extra hipcc flags]
set -e
NAME=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/lammps-ucg-dev_amd/csrc"
F="-O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wall -Wno-unused-function -Wno-unused-result"
mkdir -p "$ROOT/_ab/libs/obj_$NAME"
/opt/rocm/bin/hipcc $F "$@" -c ucg_pair_hot.hip -o "$ROOT/_ab/libs/obj_$NAME/ucg_pair_hot.o"
OBJS=$(ls *.o | grep -v "^ucg_pair_hot.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o "$ROOT/_ab/libs/libucg_$NAME.so" $OBJS "$ROOT/_ab/libs/obj_$NAME/ucg_pair_hot.o" -ldl
echo "built _ab/libs/libucg_$NAME.so"
