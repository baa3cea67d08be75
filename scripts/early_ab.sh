#!/bin/bash
# Same-box A/B of the split chain pre-pass + early execute launches (CZ_EARLY=1, the default) against one chain launch (CZ_EARLY=0):
#   scripts/early_ab.sh <out-file> [workload frames ...]      (default: mix 12500, real 16000, full_4a 10000, full_4b 10000)
set -o pipefail
OUT=$1; shift
[ $# -eq 0 ] && set -- mix 12500 real 16000 full_4a 10000 full_4b 10000 huf_literals 10000 raw_rle 10000
: > $OUT
while [ $# -ge 2 ]; do
  echo "== $1 $2 frames" >> $OUT
  for e in 1 0; do CZ_CHECK=${CZ_CHECK:-0} CZ_EARLY=$e timeout -k 10 400 python scripts/kernel_times.py $1 $2 cairo_zstd_amd/csrc/libcairo_zstd_amd.so 2>&1 | grep -v amdgpu.ids >> $OUT; done
  shift; shift
done
cut -c1-250 $OUT
