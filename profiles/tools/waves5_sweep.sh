cd /root/repo
export SLAM_HIP_LIB=$PWD/slam_amd/lib/diag/libslam_hip_w5.so
for cfg in "10240 --c3-chunks" "8960 --c3-chunks" "7680 --c3-chunks" "6400 --c3-chunks"; do
  echo "== c3 lds $cfg (5 waves)"
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --legs c3 --no-cpu --c3-walk-lds $cfg 2> gpurun_out/w5.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = d['c3']
print({k: p[k] for k in ('ms','ms_min','jobs_left_to_general_kernel','walk_lds_bytes') if k in p})
" || exit 1
done
for lds in 7680 8960 10240; do
  echo "== prod lds $lds (5 waves)"
  timeout -k 10 200 python bench.py --steps 20 --warmup 5 --legs prod --no-cpu --prod-walk-lds $lds 2> gpurun_out/w5.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = d['prod']
print({k: p[k] for k in ('ms','ms_min','jobs_left_to_general_kernel','walk_lds_bytes') if k in p}, p['one_frame']['ms'])
" || exit 1
done
