#!/bin/bash
# SQ instruction counters of the FAST kernels on the C4 shape per diagnostic setting (SLAM_HIP_FAST_DIAG bits:
# 1 = no pair test, 2 = no scoring, 4 = no suppression).  usage: profiles/tools/fast_counters.sh "0 1 7"
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
for d in ${1:-0}; do
  OUT=$PWD/gpurun_out/fast_counters_$d
  mkdir -p $OUT
  SLAM_HIP_FAST_DIAG=$d timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU \
    -d $OUT/p -o p -- python3 $PWD/profiles/tools/fast_probe.py > $OUT/p.log 2>&1
  echo "== diag $d"
  python3 profiles/tools/summarise_counters.py $OUT/p fast_tile | grep -v "^kernel"
done
