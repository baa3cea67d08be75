#!/bin/bash
set -o pipefail
L=cairo_zstd_amd/csrc/exp
for g in 2 4 6 8 12 16; do
  echo "== mix exec WGs per CU $g"
  CZ_EXEC_PER_CU=$g timeout -k 10 300 python scripts/kernel_times.py mix 12500 $L/libcz_fm6.so 2>&1 | grep -v amdgpu.ids || exit 1
done
