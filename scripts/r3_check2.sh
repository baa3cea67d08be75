#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r3/t2.log 2>&1; echo "pytest rc=$?" ; tail -3 gpurun_out/r3/t2.log
L=cairo_zstd_amd/csrc
for k in full_4a mix full_4b; do
  n=10000; [ $k = mix ] && n=12500
  timeout -k 10 600 python scripts/kernel_times.py $k $n $L/exp/libcz_nofast.so $L/exp/libcz_ew4.so $L/exp/libcz_ew5.so $L/libcairo_zstd_amd.so $L/exp/libcz_ew8.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3/kt2.log
done
