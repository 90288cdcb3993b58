#!/bin/bash
# Same-box A/B of two builds of the library on the bench's walk, C1 and C3 legs, alternating twice (boxes differ by a
# few per cent, so a comparison across gpurun calls says little).
# usage: profiles/tools/ab_bench.sh <libA.so> <libB.so>       (paths relative to the repository root)
A=${1:-slam_amd/lib/diag/libslam_hip_prev.so}; B=${2:-slam_amd/lib/libslam_hip.so}
for i in 1 2; do
for lib in $A $B; do
SLAM_HIP_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --no-cpu --no-fast 2>&1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', d['ms_per_step'], d['roofline']['kernel_ms'], d['c1']['ms'], d['c3']['ms'])
" || exit 1
done; done
