#!/bin/bash
# Soak run of the randomised differential GPU tests with every single search sent through the lean single-search
# kernel first (SLAM_HIP_SEARCH_LEAN_MIN=1) under LDS budgets that move its row-block boundaries, and of the batched
# lean-search test, on other input streams (SLAM_TEST_SEED shifts the seeds).
# usage: profiles/tools/soak2.sh [first seed] [last seed]
set -o pipefail
mkdir -p gpurun_out
for s in $(seq ${1:-1} ${2:-4}); do
  for kb in 4 6 10 33; do
    echo "seed shift $s, search LDS $kb KB"
    SLAM_TEST_SEED=$s SLAM_HIP_SEARCH_LEAN_MIN=1 SLAM_HIP_SEARCH_LDS_KB=$kb timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread -k "match_random or match_batches or match_edge or fast_random" > gpurun_out/soak2_${s}_$kb.log 2>&1 || { tail -30 gpurun_out/soak2_${s}_$kb.log | cut -c1-300; exit 1; }
    tail -1 gpurun_out/soak2_${s}_$kb.log
  done
done
