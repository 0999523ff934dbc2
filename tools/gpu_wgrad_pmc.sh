#!/bin/bash
# Counters of one weight-gradient kernel form: tools/gpu_wgrad_pmc.sh <tag> <layer> <form>   (e.g. ring enc.8 bf16t/3)
set -e
R=$GRAFT_REPO_ROOT; T=$1; export WGB_ONLY=$2 WGB_FORMS=$3
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $R/gpurun_out/${T}_pmc1 -- python3 $R/tools/wgrad_bench.py > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_ACTIVE_INST_MISC \
  --kernel-trace --output-format csv -d $R/gpurun_out/${T}_pmc2 -- python3 $R/tools/wgrad_bench.py > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
csv.field_size_limit(1 << 30)
for d in ("${T}_pmc1", "${T}_pmc2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "wgrad" in r["Kernel_Name"] and "reduce" not in r["Kernel_Name"]:
                k = r["Kernel_Name"].replace("(anonymous namespace)::", "").split("(")[0][-60:]
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                acc[k]["us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    for k, cs in acc.items():
        print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()})
PY
