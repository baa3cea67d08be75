#!/bin/bash
# round 3, first GPU check: the new execute-only kernel (parity, then A/B of its register caps)
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r3/t1.log 2>&1; echo "pytest rc=$?" ; tail -3 gpurun_out/r3/t1.log
for k in full_4a mix; do
  n=10000; [ $k = mix ] && n=12500
  timeout -k 10 600 python scripts/kernel_times.py $k $n cairo_zstd_amd/csrc/exp/libcz_ew4.so cairo_zstd_amd/csrc/exp/libcz_ew5.so cairo_zstd_amd/csrc/libcairo_zstd_amd.so cairo_zstd_amd/csrc/exp/libcz_ew8.so 2>&1 | tee -a gpurun_out/r3/kt1.log
done
CZ_EXEC=0 timeout -k 10 300 python scripts/kernel_times.py full_4a 10000 cairo_zstd_amd/csrc/libcairo_zstd_amd.so 2>&1 | tee -a gpurun_out/r3/kt1.log
