# C3 leg of the bench at several LDS budgets of the lean walk kernel, whole templates and template chunks
cd "$(dirname "$0")/../.."
for cfg in "10240" "10240 --c3-chunks" "8960 --c3-chunks" "7680 --c3-chunks" "6400 --c3-chunks"; do
  echo "== c3 lds $cfg"
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --legs c3 --cpu-seconds 1 --c3-walk-lds $cfg 2> gpurun_out/c3_sweep.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
p = d['c3']
print({k: p[k] for k in ('ms','ms_min','jobs_left_to_general_kernel','walk_lds_bytes','walk_chunks','parity') if k in p})
" || exit 1
done
