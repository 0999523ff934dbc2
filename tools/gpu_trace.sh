#!/bin/bash
# kernel-trace stats of one bench command: tools/gpu_trace.sh <tag> <bench args...>  -> gpurun_out/<tag>_kernel_stats.csv
set -e
R=$GRAFT_REPO_ROOT; T=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/${T}_trace -- python3 $R/bench.py "$@" > $R/gpurun_out/${T}_trace.json 2> $R/gpurun_out/${T}_trace.err
cd $R
f=$(find gpurun_out/${T}_trace -name "*kernel_stats.csv" | head -1); cp $f gpurun_out/${T}_kernel_stats.csv
python3 - <<PY
import csv
for r in list(csv.DictReader(open("gpurun_out/${T}_kernel_stats.csv")))[:14]:
    print(r["Name"][:90], r["Calls"], round(float(r["AverageNs"])/1e3,1), "us", r["Percentage"])
PY
