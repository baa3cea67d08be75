for cus in 96 112 128 144 160; do for lv in 4 7 10; do
  CZ_WEXEC=1,$cus,$lv bash scripts/kt_short.sh 100 sw_${cus}_${lv} full_4a 10000 cairo_zstd_amd/csrc/libcairo_zstd_amd.so | grep total | awk -v c=$cus -v l=$lv '{print c, l, "exec", $(NF-13), $(NF-12), "wexec", $(NF-16)}'
done; done
