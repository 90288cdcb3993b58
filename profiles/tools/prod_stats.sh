#!/bin/bash
# rocprofv3 --kernel-trace --stats of the bench's prod leg (whole batches and one frame's walks)
cd "$(dirname "$0")/../.."; export TMPDIR=/tmp
timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/prod_stats -o st -- python3 bench.py --no-cpu --legs prod --steps 5 --warmup 2 --repeats 2 > gpurun_out/prod_stats.log 2>&1
python3 - <<'PY'
import sqlite3, sys
sys.path.insert(0, 'profiles/tools')
import summarise_prof2 as S
print(S.kernel_stats(sqlite3.connect('gpurun_out/prod_stats/st_results.db'))[0])
PY
