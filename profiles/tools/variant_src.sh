#!/bin/bash
# Variant of the library with extra compiler flags for ONE of its sources (diagnostic A/B builds):
#   profiles/tools/variant_src.sh <source without .hip> <name> <flags...>   -> slam_amd/lib/diag/libslam_hip_<name>.so
set -e
cd "$(dirname "$0")/../.."
CS=slam_amd/csrc; LIB=slam_amd/lib
src=$1; name=$2; shift 2
mkdir -p $LIB/diag
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Iinclude -I$CS "$@" -c $CS/$src.hip -o $LIB/diag/${src}_$name.o
objs=""
for f in context pyramid match fast newpoints project; do
  if [ $f = $src ]; then objs="$objs $LIB/diag/${src}_$name.o"; else objs="$objs $LIB/obj/$f.o"; fi
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $LIB/diag/libslam_hip_$name.so $objs
rm $LIB/diag/${src}_$name.o
