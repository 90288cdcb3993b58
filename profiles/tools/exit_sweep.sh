# Instruction counters of diagnostic builds that end every wave at one phase boundary (-DSLAM_DIAG_EXIT_AT=<stamp id>,
# linked as slam_amd/lib/libslam_hip_x<id>.so): differences between consecutive boundaries give the cost of a phase.
# Code after the exit is removed by the compiler, and so is earlier work whose results only that code used.
set -e
export TMPDIR=/tmp
for id in ${EXIT_IDS:-3 45 46 44 40 41}; do
  SLAM_HIP_LIB=slam_amd/lib/libslam_hip_x$id.so timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d gpurun_out/prof_exit/x$id -o x -- python3 profiles/tools/track_probe.py coarsest > gpurun_out/prof_exit_x$id.log 2>&1
  echo "x$id done"
done
