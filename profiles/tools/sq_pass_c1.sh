#!/bin/bash
# SQ / LDS counters of the walk kernels in the bench's C1 leg.  Each --pmc pass is a run of its own.
# usage: profiles/tools/sq_pass_prod.sh <outdir-tag> [bench args...]     -> gpurun_out/sq_<tag>/<pass>/*_results.db
set -e
TAG=${1:-x}; shift || true
OUT=$PWD/gpurun_out/sq_$TAG
mkdir -p $OUT
export TMPDIR=/tmp
BENCH="python3 $PWD/bench.py --no-cpu --legs c1 --steps 3 --warmup 1 --repeats 2 $*"
timeout -s KILL 180 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES -d $OUT/insts -o insts -- $BENCH > $OUT/insts.log 2>&1
echo "insts pass done"
timeout -s KILL 180 rocprofv3 --pmc SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_INST_CYCLES_SALU SQ_WAIT_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS -d $OUT/busy -o busy -- $BENCH > $OUT/busy.log 2>&1
echo "busy pass done"
timeout -s KILL 180 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_SMEM SQ_ACTIVE_INST_ANY -d $OUT/lds -o lds -- $BENCH > $OUT/lds.log 2>&1
echo "lds pass done"
python3 $PWD/profiles/tools/summarise_counters.py $OUT
