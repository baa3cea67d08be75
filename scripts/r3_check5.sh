#!/bin/bash
# exec-kernel workgroups per CU sweep (diagnostic)
set -o pipefail
mkdir -p gpurun_out/r3
L=cairo_zstd_amd/csrc/exp
for wl in full_4a full_4b; do
for g in 4 6 8 10 12 16; do
  echo "== $wl exec WGs per CU $g" | tee -a gpurun_out/r3/kt5.log
  CZ_EXEC_PER_CU=$g timeout -k 10 300 python scripts/kernel_times.py $wl 10000 $L/libcz_nc4x.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3/kt5.log || exit 1
done
for g in 20 24 32; do
  echo "== $wl exec WGs per CU $g" | tee -a gpurun_out/r3/kt5.log
  CZ_EXEC_PER_CU=$g timeout -k 10 300 python scripts/kernel_times.py $wl 10000 $L/libcz_nc8x.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3/kt5.log || exit 1
done
done
