#!/bin/bash
# A/B of exec-kernel builds (diagnostic)
set -o pipefail
mkdir -p gpurun_out/r3
L=cairo_zstd_amd/csrc/exp
for wl in full_4a mix full_4b; do
  n=10000; [ $wl = mix ] && n=12500
  echo "== $wl" | tee -a gpurun_out/r3/kt4.log
  timeout -k 10 500 python scripts/kernel_times.py $wl $n $L/libcz_base.so $L/libcz_nc4.so $L/libcz_nc5.so $L/libcz_nc6.so $L/libcz_nc8.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3/kt4.log || exit 1
done
