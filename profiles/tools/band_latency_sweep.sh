cd /root/repo
for bl in ${BAND_LDS:-0 65536}; do for nb in ${BAND_NB:-8}; do
  echo "== band lds $bl bands $nb"
  SLAM_HIP_WALK_BAND_LDS_BYTES=$bl SLAM_HIP_WALK_BANDS=$nb timeout -k 10 200 python bench.py --steps 20 --warmup 5 --legs prod --no-cpu 2> gpurun_out/prodb.err | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1]); p = d['prod']
print(p['ms'], p['jobs_left_to_general_kernel'], p['one_frame']['ms'], p['one_frame']['ms_min'])
" || exit 1
done; done
