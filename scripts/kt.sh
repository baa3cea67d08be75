#!/bin/bash
# per-kernel times of one workload for one or more library builds: scripts/kt.sh <out-name> <workload> <frames> [lib ...]
set -o pipefail
N=$1; shift
mkdir -p gpurun_out/r4
timeout -k 10 300 python scripts/kernel_times.py "$@" > gpurun_out/r4/$N.log 2>&1; rc=$?
grep -v amdgpu.ids gpurun_out/r4/$N.log | tail -12
exit $rc
