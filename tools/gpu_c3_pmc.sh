#!/bin/bash
# counters of the first-layer forward kernels: tools/gpu_c3_pmc.sh <tag>
set -e
R=$GRAFT_REPO_ROOT; T=$1
python3 $R/tools/c3_bf16_bench.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS \
  --kernel-trace --output-format csv -d $R/gpurun_out/${T}_pmc1 -- python3 $R/tools/c3_bf16_bench.py > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INSTS_LDS SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_VMEM_WR \
  --kernel-trace --output-format csv -d $R/gpurun_out/${T}_pmc2 -- python3 $R/tools/c3_bf16_bench.py > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob, collections
csv.field_size_limit(1 << 30)
for d in ("${T}_pmc1", "${T}_pmc2"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(f"gpurun_out/{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if "c3_pkernel" in r["Kernel_Name"]:
                k = r["Kernel_Name"].split("(")[0][-40:]
                acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
                acc[k]["us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-3)
    for k, cs in acc.items():
        print(k, {c: round(sum(v) / len(v), 1) for c, v in cs.items()})
PY
