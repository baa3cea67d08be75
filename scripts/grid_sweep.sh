for g in ${GRIDS:-1 2 3 4}; do echo "grid/CU $g"; CZ_GRID_PER_CU=$g timeout -k 10 200 python scripts/kernel_times.py mix 12500 cairo_zstd_amd/csrc/exp/libcz_grid.so 2>&1 | grep total || exit 1; done
