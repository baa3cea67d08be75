mkdir -p gpurun_out/r2
timeout -k 10 500 python bench.py > gpurun_out/r2/final_bench.json 2> gpurun_out/r2/final_bench.err || { tail -5 gpurun_out/r2/final_bench.err; exit 1; }
python3 - <<'PY'
import json
d=json.loads(open("gpurun_out/r2/final_bench.json").read().strip().splitlines()[-1])
r=d["roofline"]
print(d["metric"], round(d["value"]), d["unit"], "ms", round(d["ms_per_step"],3), "frac", round(r["frac"],4), "traffic", r["traffic"], r.get("traffic_source","")[:90])
print("bit_exact", d["bit_exact"], d["frames_verified_vs_oracle_rank0"], "cpu", round(d["cpu_baseline"]["value"]))
PY
