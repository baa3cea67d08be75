#!/bin/bash
# One gpurun call's worth of checks: GPU parity suite, the pre-pass refill profile on the mix, and per-kernel times for mix and full_4a.
set -o pipefail
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py -x -q 2>&1 | tail -2 &&
CAIRO_ZSTD_AMD_LIB=cairo_zstd_amd/csrc/libcairo_zstd_amd_prof.so timeout -k 10 300 python scripts/phase_profile.py mix 12500 prepass 2>&1 | tail -3 &&
timeout -k 10 300 python scripts/mix_order_sweep.py 2>&1 | grep "index order" &&
timeout -k 10 300 python scripts/kernel_times.py full_4a 10000 cairo_zstd_amd/csrc/libcairo_zstd_amd.so 2>&1 | tail -1
