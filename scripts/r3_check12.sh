#!/bin/bash
# chain slots per wave (what more slots buy) — diagnostic builds make exp EXPFLAGS=-DCZC_SLOTS=k
set -o pipefail
L=cairo_zstd_amd/csrc
for wl in full_4a mix; do
  n=10000; [ $wl = mix ] && n=12500
  echo "== $wl $n frames"
  timeout -k 10 500 python scripts/kernel_times.py $wl $n $L/exp/libcz_sl8.so $L/exp/libcz_sl9.so $L/libcairo_zstd_amd.so $L/exp/libcz_sl11.so $L/exp/libcz_sl12.so 2>&1 | grep -v amdgpu.ids || exit 1
done
echo "== full_4a 20000 frames (two rounds of slots)"
timeout -k 10 500 python scripts/kernel_times.py full_4a 20000 $L/libcairo_zstd_amd.so $L/exp/libcz_sl11.so $L/exp/libcz_sl12.so 2>&1 | grep -v amdgpu.ids
