#!/bin/bash
# LDS-array counters of the FAST tile kernel on the C4 shape per diagnostic setting (SLAM_HIP_FAST_DIAG bits: 1 = no pair
# test, 2 = no scoring, 4 = no suppression): which phase the bank-conflict cycles belong to.  usage: fast_lds_counters.sh "0 1 3 7"
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
for d in ${1:-0}; do
  OUT=$PWD/gpurun_out/fast_lds_$d
  rm -rf $OUT; mkdir -p $OUT
  SLAM_HIP_FAST_DIAG=$d timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU \
    -d $OUT/p -o p -- python3 $PWD/profiles/tools/fast_probe.py > $OUT/p.log 2>&1
  echo "== diag $d"
  python3 profiles/tools/summarise_counters.py $OUT/p fast_tile | grep -v "^kernel"
  rm -rf $OUT
done
