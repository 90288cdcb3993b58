#!/bin/bash
# Per-phase instruction counts of the lean walk kernel: one diagnostic library per (level, phase boundary) whose waves
# end there (built by `profiles/tools/exit_sweep2.sh build` on the CPU), then one SQ counter pass per library on the GPU
# (`profiles/tools/exit_sweep2.sh run`).  Differences between consecutive rows are the phases' costs.
set -e
cd "$(dirname "$0")/../.."
CS=slam_amd/csrc
LIB=slam_amd/lib
POINTS=${EXIT_SWEEP_POINTS:-"2:2 2:50 2:51 2:52 2:53 2:54 2:55 1:2 0:2 0:50 0:51 0:52 0:53 0:54 0:55 -1:30"}
if [ "$1" = build ]; then
  mkdir -p $LIB/diag
  n=0
  for p in $POINTS; do
    lvl=${p%%:*}; id=${p##*:}
    ( /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -Iinclude -I$CS \
        -DSLAM_DIAG_EXIT_AT=$id -DSLAM_DIAG_EXIT_LEVEL=$lvl -c $CS/match.hip -o $LIB/diag/match_${lvl}_$id.o &&
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $LIB/diag/libslam_hip_exit_${lvl}_$id.so \
        $LIB/obj/context.o $LIB/obj/pyramid.o $LIB/obj/fast.o $LIB/obj/newpoints.o $LIB/obj/project.o $LIB/diag/match_${lvl}_$id.o &&
      rm $LIB/diag/match_${lvl}_$id.o ) &
    n=$((n+1)); if [ $((n % 8)) = 0 ]; then wait; fi
  done
  wait
  ls $LIB/diag
else
  export TMPDIR=/tmp
  OUT=$PWD/gpurun_out/exit_sweep2
  mkdir -p $OUT
  for p in $POINTS; do
    lvl=${p%%:*}; id=${p##*:}
    SLAM_HIP_LIB=$PWD/$LIB/diag/libslam_hip_exit_${lvl}_$id.so timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVES \
      -d $OUT/p_${lvl}_$id -o p -- python3 $PWD/profiles/tools/walk_probe.py 128 2 > $OUT/p_${lvl}_$id.log 2>&1
    if [ -n "$EXIT_SWEEP_LDS" ]; then
      SLAM_HIP_LIB=$PWD/$LIB/diag/libslam_hip_exit_${lvl}_$id.so timeout -s KILL 120 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU \
        -d $OUT/p_${lvl}_$id/lds -o q -- python3 $PWD/profiles/tools/walk_probe.py 128 2 > $OUT/q_${lvl}_$id.log 2>&1
    fi
    echo "level $lvl stamp $id"
    python3 profiles/tools/summarise_counters.py $OUT/p_${lvl}_$id track_lean | grep -v "^kernel" || true
  done
fi
