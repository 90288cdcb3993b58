#!/bin/bash
# Builds a variant of the library with extra compiler flags for match.hip (diagnostic A/B builds):
#   profiles/tools/variant.sh build <name> <flags...>   -> slam_amd/lib/diag/libslam_hip_<name>.so
#   profiles/tools/variant.sh count <name>...           -> instruction counters of the lean walk kernel per variant (on the GPU)
set -e
cd "$(dirname "$0")/../.."
CS=slam_amd/csrc; LIB=slam_amd/lib
cmd=$1; shift
if [ "$cmd" = build ]; then
  name=$1; shift
  mkdir -p $LIB/diag
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Iinclude -I$CS "$@" -c $CS/match.hip -o $LIB/diag/match_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $LIB/diag/libslam_hip_$name.so \
     $LIB/obj/context.o $LIB/obj/pyramid.o $LIB/obj/fast.o $LIB/obj/newpoints.o $LIB/obj/project.o $LIB/diag/match_$name.o
  rm $LIB/diag/match_$name.o
else
  for name in "$@"; do
    echo "== $name"
    SLAM_HIP_LIB=$PWD/$LIB/diag/libslam_hip_$name.so profiles/tools/walk_counters.sh $name | grep -E "VALU|SALU|INSTS_LDS"
  done
fi
