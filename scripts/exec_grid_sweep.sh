#!/bin/bash
# exec_frames alone (no wexec) at 6/8/10/12/16 waves per CU; and nt record loads; and side by side
O=gpurun_out/r5/exec_grid_sweep.txt; mkdir -p gpurun_out/r5; : > $O
for g in 6 8 10 12 16; do echo "== exec only, $g waves per CU" >> $O; CZ_WEXEC=0 CZ_EXEC_PER_CU=$g timeout -k 10 300 python scripts/kernel_times.py full_4a 10000 cairo_zstd_amd/csrc/exp/libcz_x.so 2>&1 | grep -v amdgpu.ids | cut -c1-170 >> $O; done
for g in 8 16; do echo "== exec only, nt record loads, $g waves per CU" >> $O; CZ_WEXEC=0 CZ_EXEC_PER_CU=$g timeout -k 10 300 python scripts/kernel_times.py full_4a 10000 cairo_zstd_amd/csrc/exp/libcz_ntrec.so 2>&1 | grep -v amdgpu.ids | cut -c1-170 >> $O; done
echo "== side by side (product arrangement), plain / nt record loads" >> $O
timeout -k 10 300 python scripts/kernel_times.py full_4a 10000 cairo_zstd_amd/csrc/exp/libcz_x.so cairo_zstd_amd/csrc/exp/libcz_ntrec.so 2>&1 | grep -v amdgpu.ids | cut -c1-170 >> $O
echo "== real, plain / nt" >> $O
timeout -k 10 300 python scripts/kernel_times.py real 16000 cairo_zstd_amd/csrc/exp/libcz_x.so cairo_zstd_amd/csrc/exp/libcz_ntrec.so 2>&1 | grep -v amdgpu.ids | cut -c1-170 >> $O
cat $O
