#!/bin/bash
# cz_huf1_kernel workgroups per CU beside the chain kernel (diagnostic)
set -o pipefail
L=cairo_zstd_amd/csrc/exp
for wl in full_4a mix; do
n=10000; [ $wl = mix ] && n=12500
for g in 1 2 3 4 5 6; do
  echo "== $wl huf1 WGs per CU $g"
  CZ_HUF1_PER_CU=$g timeout -k 10 300 python scripts/kernel_times.py $wl $n $L/libcz_h1.so 2>&1 | grep -v amdgpu.ids || exit 1
done
done
