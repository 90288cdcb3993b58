cd /root/repo
for lib in "" "$PWD/slam_amd/lib/diag/libslam_hip_box1k.so"; do
for cfg in "10240 --c3-chunks" "11520 --c3-chunks"; do
  echo "== lib=$lib c3 $cfg"
  SLAM_HIP_LIB=$lib timeout -k 10 200 python bench.py --steps 20 --warmup 5 --legs c3 --cpu-seconds 1 --c3-walk-lds $cfg 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = d['c3']
print(p['ms'], p['ms_min'], p['jobs_left_to_general_kernel'], p['parity'][:20])
"
done; done
