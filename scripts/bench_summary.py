import json,sys
d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("4a", d["ms_per_step"], d["roofline"]["frac"], d.get("exec_stage"))
for k,v in (d.get("other_workloads") or {}).items():
    print(k, v.get("ms_per_step"), {kk:v[kk] for kk in v if "ms" in kk and kk!="ms_per_step"})
