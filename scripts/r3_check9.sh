#!/bin/bash
set -o pipefail
L=cairo_zstd_amd/csrc/exp
for wl in mix full_4a full_4b; do
  n=10000; [ $wl = mix ] && n=12500
  echo "== $wl"
  timeout -k 10 500 python scripts/kernel_times.py $wl $n $L/libcz_fm6.so $L/libcz_calm4.so $L/libcz_calm8.so $L/libcz_calm16.so 2>&1 | grep -v amdgpu.ids || exit 1
done
CAIRO_ZSTD_AMD_LIB=$PWD/$L/libcz_calm8.so timeout -k 10 300 python scripts/mix_each.py 12 2>&1 | grep -v amdgpu.ids
