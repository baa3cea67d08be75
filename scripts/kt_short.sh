#!/bin/bash
# like kt.sh with a short leash (hang hunting): scripts/kt_short.sh <seconds> <out-name> <workload> <frames> [lib ...]
T=$1; N=$2; shift; shift
mkdir -p gpurun_out/r4
timeout -k 5 $T python -u scripts/kernel_times.py "$@" > gpurun_out/r4/$N.log 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/r4/$N.log | tail -4
echo "rc=$rc"
exit 0
