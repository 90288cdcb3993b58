#!/bin/bash
# Time of track_band_kernel (one frame's walks of the bench's prod leg, 8 bands each) with waves that end at a chosen stamp
# of the last level: builds from `profiles/tools/variant.sh build exit_<level>_<stamp> -DSLAM_DIAG_EXIT_AT=.. -DSLAM_DIAG_EXIT_LEVEL=..`
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp
for v in exit_0_2 exit_0_50 exit_0_52 exit_0_53 exit_0_54 full; do
  lib=$PWD/slam_amd/lib/diag/libslam_hip_$v.so; [ $v = full ] && lib=$PWD/slam_amd/lib/libslam_hip.so
  rm -rf gpurun_out/bp_$v
  SLAM_HIP_LIB=$lib timeout -k 10 120 rocprofv3 --kernel-trace --stats -d gpurun_out/bp_$v -o st -- python3 bench.py --no-cpu --legs prod --steps 3 --warmup 1 --repeats 2 > gpurun_out/bp_$v.log 2>&1
  python3 - "$v" <<'PY'
import sqlite3, sys
db = sqlite3.connect('gpurun_out/bp_%s/st_results.db' % sys.argv[1])
for n, c, a, lo in db.execute("select name, count(*), avg(duration), min(duration) from kernels where name like '%track_band_kernel%' group by name"):
    print(sys.argv[1], 'track_band_kernel', c, 'avg_us %.1f' % (a / 1e3), 'min_us %.1f' % (lo / 1e3))
PY
  rm -rf gpurun_out/bp_$v
done
