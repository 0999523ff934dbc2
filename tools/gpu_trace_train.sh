#!/bin/bash
# kernel-trace stats of the native training step: tools/gpu_trace_train.sh <tag> <precision>
set -e
R=$GRAFT_REPO_ROOT; T=$1; P=${2:-bf16}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_trace_train_$P -- python3 $R/tools/train_bench.py --clips 32 --precision $P > $R/gpurun_out/${T}_trace_train_$P.json 2> $R/gpurun_out/${T}_trace_train_$P.err
cd $R
f=$(find gpurun_out/${T}_trace_train_$P -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/${T}_train_step_${P}_kernel_stats.csv
python3 - <<PY
import csv
rows=list(csv.DictReader(open("gpurun_out/${T}_train_step_${P}_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:16]:
    print(r["Name"][:84], r["Calls"], round(float(r["TotalDurationNs"])/1e6,2), "ms", round(100*float(r["TotalDurationNs"])/tot,1))
PY
