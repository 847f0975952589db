import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import csv, glob, sys, json, collections
out = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"gpurun_out/pmc_{name}/*/*counter_collection.csv")
    if not f:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] == name:
            acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {})[name] = {"launches": len(v), "mean": sum(v) / len(v)}
sel = {k: v for k, v in out.items() if any(s in k for s in ("job_scan", "bytemap_pack", "job_index", "frame_remap", "mbk_assign", "eps_components"))}
print(json.dumps(sel, indent=1))
json.dump(sel, open("gpurun_out/pmc_summary.json", "w"), indent=1)
