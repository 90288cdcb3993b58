#!/bin/bash
# static count of scalar-spill moves (v_readlane / v_writelane with a literal lane) and register use of the lean walk kernel
cd "$(dirname "$0")/../../slam_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include -I. "$@" -S --cuda-device-only -o /tmp/match_sc.s match.hip 2>/dev/null
awk '/^_ZN12_GLOBAL__N_117track_lean_kernel/{f=1} f{print} f&&/\.amdhsa_kernel/{exit}' /tmp/match_sc.s > /tmp/tl_sc.s
echo "readlane $(grep -c 'v_readlane_b32 s[0-9]*, v[0-9]*, [0-9]' /tmp/tl_sc.s) writelane $(grep -c 'v_writelane_b32' /tmp/tl_sc.s) lines $(wc -l < /tmp/tl_sc.s)"
grep -A30 "amdhsa_kernel _ZN12_GLOBAL__N_117track_lean_kernel" /tmp/match_sc.s | grep -E "next_free_vgpr|next_free_sgpr"
