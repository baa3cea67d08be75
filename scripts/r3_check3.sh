#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/r3
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py -x -q > gpurun_out/r3/t3.log 2>&1; echo "pytest rc=$?" ; tail -3 gpurun_out/r3/t3.log
L=cairo_zstd_amd/csrc
timeout -k 10 600 python scripts/kernel_times.py full_4a 10000 $L/exp/libcz_ew4.so $L/exp/libcz_ew5.so $L/libcairo_zstd_amd.so $L/exp/libcz_ew8.so 2>&1 | grep -v amdgpu.ids | tee -a gpurun_out/r3/kt3.log
CAIRO_ZSTD_AMD_LIB=$PWD/$L/exp/libcz_ew4.so bash scripts/sq_counters.sh full_4a 10000 2>&1 | grep -v amdgpu.ids
