#!/bin/bash
# rocprofv3 --kernel-trace --stats of the driver's own bench command (--steps 20 --warmup 5), main leg only
# usage: profiles/tools/driver_cmd_stats.sh <tag>   -> gpurun_out/prof_<tag>/driver/
set -e
TAG=${1:-r03}
OUT=$PWD/gpurun_out/prof_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
timeout -k 10 240 rocprofv3 --kernel-trace --stats -d $OUT/driver -o stats -- python3 $PWD/bench.py --steps 20 --warmup 5 --no-cpu --legs none > $OUT/driver.log 2>&1
echo "driver-command pass done"
