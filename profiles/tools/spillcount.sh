#!/bin/bash
# static count of scalar-spill moves (v_readlane / v_writelane with a literal lane) and register use of the two builds of
# the lean walk kernel (track_lean_kernel<false>: whole levels; <true>: blocks of cell rows and chunks of template rows)
cd "$(dirname "$0")/../../slam_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -I../../include -I. "$@" -S --cuda-device-only -o /tmp/match_sc.s match.hip 2>/dev/null
for b in 0 1; do
  sym=$(grep -o "^_ZN12_GLOBAL__N_117track_lean_kernelILb${b}E[A-Za-z0-9_]*" /tmp/match_sc.s | head -1)
  awk -v s="$sym:" '$1==s{f=1} f{print} f&&/s_endpgm/{exit}' /tmp/match_sc.s > /tmp/tl_sc_$b.s
  echo "build $b: readlane $(grep -c 'v_readlane_b32 s[0-9]*, v[0-9]*, [0-9]' /tmp/tl_sc_$b.s) writelane $(grep -c 'v_writelane_b32' /tmp/tl_sc_$b.s) lines $(wc -l < /tmp/tl_sc_$b.s)" \
       $(grep -E "^\s*\.set $sym\.(num_vgpr|numbered_sgpr|private_seg_size)" /tmp/match_sc.s | sed 's/.*\.\([a-z_]*, [0-9]*\)$/\1;/' | tr '\n' ' ')
done
