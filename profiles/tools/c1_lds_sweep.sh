#!/bin/bash
# C1 leg of the bench (16384 single searches) over LDS budgets of the lean single-search kernel.
for k in ${*:-6 8 10 12}; do SLAM_HIP_SEARCH_LDS_KB=$k timeout -k 10 300 python3 bench.py --no-cpu --no-fast 2>&1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print($k, d['c1']['ms'], d['ms_per_step'], d['c3']['ms'])
" || exit 1; done
