#!/bin/bash
# Profiles the default bench command on the GPU box.  Outputs under gpurun_out/prof_<tag>/:
#   stats/   rocprofv3 --kernel-trace --stats of `python3 bench.py --no-cpu`
#   fetch/   --pmc FETCH_SIZE pass, write/  --pmc WRITE_SIZE pass (separate runs: TCC slots)
#   lds/     --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE (LDS array cycles and the clock they ran at)
#   calib_*/ the same two counters on kernels with known byte counts (profiles/tools/pmc_calib.hip)
# usage: profiles/tools/profile_bench.sh <tag>
set -e
TAG=${1:-r02}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --steps 20 --warmup 3 --repeats 5 --no-cpu --no-fast-sparse"
BENCH_PMC="python3 $PWD/bench.py --steps 3 --warmup 1 --repeats 2 --no-cpu --no-fast-sparse"
timeout -k 10 300 rocprofv3 --kernel-trace --stats -d $OUT/stats -o stats -- $BENCH > $OUT/stats.log 2>&1
echo "stats pass done"
timeout -s KILL 240 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch -o fetch -- $BENCH_PMC > $OUT/fetch.log 2>&1
echo "fetch pass done"
timeout -s KILL 240 rocprofv3 --pmc WRITE_SIZE -d $OUT/write -o write -- $BENCH_PMC > $OUT/write.log 2>&1
echo "write pass done"
timeout -s KILL 240 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS GRBM_GUI_ACTIVE -d $OUT/lds -o lds -- $BENCH_PMC > $OUT/lds.log 2>&1
echo "lds pass done"
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $OUT/calib_fetch -o calib -- $PWD/profiles/tools/pmc_calib.bin > $OUT/calib_fetch.log 2>&1
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $OUT/calib_write -o calib -- $PWD/profiles/tools/pmc_calib.bin > $OUT/calib_write.log 2>&1
echo "calibration passes done"
find $OUT -name '*.csv' | sort
