#!/bin/bash
# Soak run of the randomised differential GPU tests on other input streams (SLAM_TEST_SEED shifts their seeds).
# usage: profiles/tools/soak.sh [first seed] [last seed]
set -o pipefail
for s in $(seq ${1:-1} ${2:-6}); do
  echo "seed shift $s"
  SLAM_TEST_SEED=$s timeout -k 10 300 python -u -m pytest tests -m gpu -x -q --timeout 200 --timeout-method thread -k "random or ellipse_setup or wide_templates or band_count or lean_coarse or chunk or graph or no_alloc" > gpurun_out/soak_$s.log 2>&1 || { tail -30 gpurun_out/soak_$s.log | cut -c1-300; exit 1; }
  tail -1 gpurun_out/soak_$s.log
done
