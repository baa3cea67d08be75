#!/bin/bash
set -o pipefail
L=cairo_zstd_amd/csrc/exp
for wl in mix full_4a; do
  n=10000; [ $wl = mix ] && n=12500
  echo "== $wl"
  timeout -k 10 500 python scripts/kernel_times.py $wl $n $L/libcz_base.so $L/libcz_fm0.so $L/libcz_fm2.so $L/libcz_fm6.so $L/libcz_fm16.so 2>&1 | grep -v amdgpu.ids || exit 1
done
