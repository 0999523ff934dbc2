#!/bin/bash
# kernel-trace timeline of one small video call: tools/gpu_trace_small.sh <tag> <bench args...>
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf $O/${T}_tr
rocprofv3 --kernel-trace --output-format csv -d $O/${T}_tr -- python3 $R/bench.py --no-split --no-cpu-baseline --no-train --no-layer-events --steps 30 --warmup 10 "$@" > $O/${T}_tr.json 2> $O/${T}_tr.err || { tail -20 $O/${T}_tr.err; exit 1; }
python3 $R/tools/trace_gaps.py $O/${T}_tr > $O/${T}.gaps.txt
python3 $R/tools/trace_timeline.py $O/${T}_tr > $O/${T}.timeline.txt
cat $O/${T}.gaps.txt; cat $O/${T}.timeline.txt
