cd /root/repo; export TMPDIR=/tmp
SLAM_HIP_LINE_LEAN_LDS_BYTES=12800 timeout -k 10 200 rocprofv3 --kernel-trace --stats -d gpurun_out/line_stats -o st -- python3 bench.py --no-cpu --legs line --steps 5 --warmup 2 --repeats 2 > gpurun_out/line_stats.log 2>&1
python3 - <<'PY'
import csv,glob
f=glob.glob('/root/repo/gpurun_out/line_stats/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'line' in r['Name']: print(r['Name'][:40], r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'])
PY
