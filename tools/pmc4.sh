#!/bin/bash
# profiles of round 4 (run on the GPU box from the repository root): kernel stats of the default bench command, PMC traffic of the
# per-pixel passes (separate FETCH_SIZE / WRITE_SIZE passes on a three-frame run), the bench line itself
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4p
python3 bench.py --steps 10 --warmup 2 > gpurun_out/r4p/bench4k.json 2> gpurun_out/r4p/bench4k.err &&
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r4p/stats -o s -- python3 bench.py > gpurun_out/r4p/stats.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d gpurun_out/r4p/pmc_FETCH_SIZE -o p -- python3 tools/oneframe.py > gpurun_out/r4p/pmc_f.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d gpurun_out/r4p/pmc_WRITE_SIZE -o p -- python3 tools/oneframe.py > gpurun_out/r4p/pmc_w.log 2>&1
python3 - <<'PY'
import csv, glob, json, collections
out = {}
for name in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"gpurun_out/r4p/pmc_{name}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == name:
                acc[r["Kernel_Name"].split("(")[0]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            out.setdefault(k, {})[name] = {"launches": len(v), "mean": sum(v) / len(v)}
sel = {k: v for k, v in out.items() if any(s in k for s in ("job_scan", "bytemap_pack", "job_index", "frame_remap", "mbk_assign", "cluster_sums", "mbk_init3"))}
json.dump(sel, open("gpurun_out/r4p/pmc_fetch_write_kb.json", "w"), indent=1)
print(json.dumps(sel, indent=1))
PY
