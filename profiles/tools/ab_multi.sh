#!/bin/bash
# Same-box comparison of several library builds on the bench's walk, C1 and C3 legs, two rounds, alternating.
# usage: profiles/tools/ab_multi.sh <lib.so>...      (paths relative to the repository root)
for i in 1 2; do
for lib in "$@"; do
SLAM_HIP_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --no-cpu --no-fast 2>&1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', round(d['ms_per_step'],4), round(d['roofline']['kernel_ms'],4), round(d['c1']['ms'],4), round(d['c3']['ms'],4), round(d['line_walk']['ms'],4))
" || exit 1
done; done
