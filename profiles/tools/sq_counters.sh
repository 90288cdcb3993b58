#!/bin/bash
# Instruction mix of the search kernel: SQ counters of one never-stopping C2 launch set (profiles/tools/track_probe.py one).
# Output: gpurun_out/prof_sq/<pass>/..._results.db; summarise with profiles/tools/summarise_sq.py.
set -e
OUT=$PWD/gpurun_out/prof_sq
mkdir -p $OUT
export TMPDIR=/tmp
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $OUT/insts -o insts -- python3 $PWD/profiles/tools/track_probe.py one > $OUT/insts.log 2>&1
echo "insts pass done"
timeout -s KILL 120 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU -d $OUT/busy -o busy -- python3 $PWD/profiles/tools/track_probe.py one > $OUT/busy.log 2>&1
echo "busy pass done"
find $OUT -name '*.db'
