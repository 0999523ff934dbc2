#!/bin/bash
# Executed-instruction mix of the bench's kernels (run on the GPU box): one rocprofv3 --pmc pass (kernel-trace only), printed per
# kernel as instructions per wave.  usage: tools/pmc_insts.sh <tag> [bench.py args...]
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_insts_$tag
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_insts_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-layer-events --no-split --no-train --no-video --no-small --stream-frames 0 "$@" > /dev/null 2>&1
cd $R
python3 - <<PY
import csv, glob
from collections import defaultdict
f = glob.glob("gpurun_out/pmc_insts_$tag/**/*counter_collection.csv", recursive=True)[0]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].split("(")[0][:60]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k, c in acc.items():
    w = c.get("SQ_WAVES", 0) or 1
    print(f"{k:60s} waves {int(w):8d}  per wave: " + "  ".join(f"{n_[9:]} {c.get(n_, 0) / w:8.1f}" for n_ in ("SQ_INSTS_VALU", "SQ_INSTS_MFMA", "SQ_INSTS_SALU", "SQ_INSTS_LDS", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "SQ_INSTS_SMEM")))
PY
