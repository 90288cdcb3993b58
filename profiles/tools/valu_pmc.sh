#!/bin/bash
# Is the chip's clock really lower under dense VALU load, or does s_memtime not count shader cycles?  One counter pass
# over profiles/tools/valu_rate2.bin (the binary directly after `--`): GRBM_GUI_ACTIVE counts graphics-clock cycles
# while the kernel runs, so GRBM cycles / kernel duration is the clock the chip held; SQ_BUSY_CYCLES and
# SQ_ACTIVE_INST_VALU say how much of that the vector ALUs were issuing.  usage: profiles/tools/valu_pmc.sh
cd "$(dirname "$0")/../.."
export TMPDIR=/tmp
OUT=$PWD/gpurun_out/valu_pmc
rm -rf $OUT; mkdir -p $OUT
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace -d $OUT/p -o p -- $PWD/profiles/tools/valu_rate2.bin > $OUT/run.log 2>&1
python3 profiles/tools/summarise_valu_pmc.py $OUT/p
