#!/bin/bash
# GPU parity suite, smoke() and the default bench line in one call: scripts/gpu_suite.sh <out-subdir> [bench args...]
set -o pipefail
D=gpurun_out/${1:-r5}; shift
mkdir -p $D
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $D/pytest_gpu.log 2>&1; rc=$?; tail -3 $D/pytest_gpu.log; [ $rc = 0 ] || exit $rc
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | grep -v amdgpu.ids | tail -2 || exit 1
timeout -k 10 600 python bench.py "$@" > $D/bench_default.json 2> $D/bench_default.err || { tail -5 $D/bench_default.err; exit 1; }
python3 - $D <<'PY'
import json, sys
d = json.loads(open(sys.argv[1] + "/bench_default.json").read().strip().splitlines()[-1])
print({k: d[k] for k in ("value", "ms_per_step", "bit_exact", "kernel_source_hash")}, d["roofline"]["frac"], d["roofline"]["traffic"])
for k, o in (d.get("other_workloads") or {}).items(): print("  ", k, o.get("kernel_ms_mean"), o.get("roofline_frac"), o.get("bit_exact"))
PY
