#!/bin/bash
# main decode kernel time against workgroups per CU (CZ_GRID_PER_CU caps the main launch only)
for g in 4 6 8 10 12 15; do
  echo "grid per CU $g"
  CZ_GRID_PER_CU=$g timeout -k 10 200 python scripts/kernel_times.py ${1:-full_4a} ${2:-10000} cairo_zstd_amd/csrc/libcairo_zstd_amd.so || exit 1
done
