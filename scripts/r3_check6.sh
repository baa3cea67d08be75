#!/bin/bash
set -o pipefail
export CAIRO_ZSTD_AMD_LIB=$PWD/cairo_zstd_amd/csrc/exp/libcz_ovl.so
for g in 0 4 6 8; do
  echo "== overlap $g WG/CU"
  if [ $g = 0 ]; then bash scripts/ktimeline.sh full_4a 10000 ovl$g || exit 1; else CZ_EXP_OVERLAP=$g bash scripts/ktimeline.sh full_4a 10000 ovl$g || exit 1; fi
done
