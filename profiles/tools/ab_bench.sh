for i in 1 2; do
for lib in slam_amd/lib/diag/libslam_hip_prev.so slam_amd/lib/libslam_hip.so; do
SLAM_HIP_LIB=$PWD/$lib timeout -k 10 300 python3 bench.py --no-cpu --no-fast 2>&1 | python3 -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('$lib', d['ms_per_step'], d['roofline']['kernel_ms'], d['c1']['ms'], d['c3']['ms'])
" || exit 1
done; done
