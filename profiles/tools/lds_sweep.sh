set -e
for l in 5120 6144 7168 8192 9216 10240 11264; do
  echo "LDS $l"
  SLAM_PROBE_LDS=$l timeout -k 10 120 python3 -u profiles/tools/track_probe.py one 2>/dev/null
  SLAM_PROBE_LDS=$l timeout -k 10 120 python3 -u profiles/tools/track_probe.py size 81 2>/dev/null
done
