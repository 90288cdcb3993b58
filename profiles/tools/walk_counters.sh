#!/bin/bash
# VALU / SALU / LDS instruction counts per walk of the lean walk kernel on the bench's workload, for the library in
# place (or SLAM_HIP_LIB).  usage: profiles/tools/walk_counters.sh [tag]
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/walk_counters_${1:-x}
mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES \
  -d $OUT/p -o p -- python3 $PWD/profiles/tools/walk_probe.py 128 2 > $OUT/p.log 2>&1
python3 profiles/tools/summarise_counters.py $OUT/p track_lean | grep -v "^kernel"
