#!/bin/bash
# LDS-array and bank-conflict cycles of the lean walk kernel up to a stamp of its last level (builds of
# `EXIT_SWEEP_POINTS="0:52 0:53 0:54 0:55 -1:30" profiles/tools/exit_sweep2.sh build`): cumulative per walk; differences
# between rows are the phases' shares (52 -> 53: tile staging; 53 -> 54: the sums, window reads and flushes; 54 -> 55: scores).
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp
for p in 0:52 0:53 0:54 0:55 -1:30; do
  lvl=${p%%:*}; id=${p##*:}
  OUT=$PWD/gpurun_out/ldsp_${lvl}_$id; rm -rf $OUT
  SLAM_HIP_LIB=$PWD/slam_amd/lib/diag/libslam_hip_exit_${lvl}_$id.so timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAVES \
    -d $OUT -o q -- python3 $PWD/profiles/tools/walk_probe.py 128 2 > $OUT.log 2>&1
  echo "level $lvl stamp $id: $(python3 profiles/tools/summarise_counters.py $OUT track_lean | grep -v '^kernel' | awk '{printf "%s=%s ", $(NF-3), $NF}')"
  rm -rf $OUT
done
