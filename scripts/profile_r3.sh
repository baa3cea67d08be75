#!/bin/bash
# Run on the GPU box (gpurun): the bench line, the rocprofv3 kernel-trace summary of the same command, and the two PMC passes
# (FETCH_SIZE, WRITE_SIZE: separate runs, never combined with a trace domain) -> gpurun_out/r3/.  usage: profile_r3.sh [workload]
set -o pipefail
WL=${1:-full_4a}
O=gpurun_out/r3
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
timeout -k 10 500 python3 bench.py --workload $WL > $O/bench_$WL.json 2> $O/bench_$WL.err &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt_$WL -- python3 bench.py --workload $WL --no-cpu-baseline --no-other-workloads > $O/bench_${WL}_under_rocprof.json 2> $O/kt_$WL.err &&
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch_$WL -- python3 bench.py --workload $WL --no-cpu-baseline --no-other-workloads --steps 3 --warmup 1 > $O/pmc_fetch_$WL.json 2> $O/pmc_fetch_$WL.err &&
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write_$WL -- python3 bench.py --workload $WL --no-cpu-baseline --no-other-workloads --steps 3 --warmup 1 > $O/pmc_write_$WL.json 2> $O/pmc_write_$WL.err &&
python3 scripts/pmc_summary.py $O/pmc_fetch_$WL $O/pmc_write_$WL $O/bench_${WL}_under_rocprof.json $O/pmc_hbm_traffic_$WL.json > /dev/null &&
find $O/kt_$WL -name "*kernel_stats.csv" -exec cp {} $O/${WL}_kernel_stats.csv \; &&
rm -rf $O/pmc_fetch_$WL/*/*.db $O/pmc_write_$WL/*/*.db $O/kt_$WL/*/*.db &&
find $O/kt_$WL -name "*kernel_trace.csv" -delete &&
tail -c 600 $O/bench_$WL.json && cat $O/${WL}_kernel_stats.csv
