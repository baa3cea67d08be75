#!/bin/bash
# cz_wexec_kernel's share of the chip side by side (config 4a): CUs x frames per CU that cz_execute_frames_kernel leaves to it at the end.
#   scripts/wexec_sweep_r5.sh "<cus list>" "<leave list>" [repeats]     -> one line per setting and repeat
for cus in ${1:-64 96 112 120 128 136 144 160}; do for lv in ${2:-3 7 12}; do for rep in $(seq 1 ${3:-1}); do
  echo -n "cus $cus leave $lv: "
  CZ_WEXEC=1,$cus,$lv timeout -k 10 120 python scripts/kernel_times.py full_4a 10000 cairo_zstd_amd/csrc/libcairo_zstd_amd.so 2>&1 | grep total | sed -e 's/.*total/total/' | cut -c1-150
done; done; done
