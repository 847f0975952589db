#!/bin/bash
# diagnostic only: SQ counters of the k-means++ chain kernel (two passes of 8 counters)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace --output-format csv -d gpurun_out/pmc_init_a -- python3 tools/initrun.py > gpurun_out/pmc_init_a.log 2>&1 &&
rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU --kernel-trace --output-format csv -d gpurun_out/pmc_init_b -- python3 tools/initrun.py > gpurun_out/pmc_init_b.log 2>&1
python3 - <<'PY'
import csv, glob, collections
for d in ("a", "b"):
    for f in glob.glob(f"gpurun_out/pmc_init_{d}/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "mbk_init2" in r["Kernel_Name"]:
                acc[r["Counter_Name"]] += float(r["Counter_Value"])
        for k, v in acc.items():
            print(d, k, v)
PY
