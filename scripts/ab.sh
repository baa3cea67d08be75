#!/bin/bash
# Same-box A/B of library builds:  scripts/ab.sh <workload> <frames> [name=FLAGS ...]
#   builds csrc/exp/libcz_<name>.so with EXPFLAGS=FLAGS for every pair (here, before the gpurun call: `make -C cairo_zstd_amd/csrc exp
#   NAME=<name> EXPFLAGS="<flags>"`), then scripts/kernel_times.py on the product library and on every csrc/exp/libcz_*.so named.
#   Environment knobs of kernel_times.py: CZ_WEXEC=on[,cus[,leave_per_cu[,force]]]  CZ_EXEC=0|1|4|8  CZ_CHECK=1 (every frame vs the oracle)
set -o pipefail
WL=$1; N=$2; shift; shift
LIBS="cairo_zstd_amd/csrc/libcairo_zstd_amd.so"
for nm in "$@"; do LIBS="$LIBS cairo_zstd_amd/csrc/exp/libcz_${nm%%=*}.so"; done
timeout -k 10 500 python scripts/kernel_times.py $WL $N $LIBS 2>&1 | grep -v amdgpu.ids
