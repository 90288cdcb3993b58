set -e
cd /root/repo
for lds in ${PROD_LDS:-10240 11520 12800 15360}; do
  echo "== prod lds $lds"
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --legs prod --cpu-seconds 1 --prod-walk-lds $lds 2> gpurun_out/prod_$lds.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
p = d['prod']
print({k: p[k] for k in ('ms','ms_min','jobs_left_to_general_kernel','walk_lds_bytes','frac_of_packed_dot_peak','parity') if k in p}, p['one_frame']['ms'], d['roofline']['kernel_ms'])
"
done
