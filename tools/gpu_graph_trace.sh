#!/bin/bash
# Eager launches vs hipGraph replay at the reference's call sizes, from kernel traces:
# for each size one `rocprofv3 --kernel-trace` run eager and one with --graph, summarised by tools/trace_gaps.py
# (span / busy / idle per step, per-kernel sums) and tools/trace_timeline.py (one step's launches with queue ids).
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r03g}
C="--no-split --no-cpu-baseline --no-train --no-layer-events --steps 30 --warmup 10"
cd /tmp && export TMPDIR=/tmp
run() {   # tag, bench args...
  local tag=$1; shift
  rm -rf $O/${T}_$tag
  rocprofv3 --kernel-trace --output-format csv -d $O/${T}_$tag -- python3 $R/bench.py $C "$@" > $O/${T}_$tag.json 2> $O/${T}_$tag.err || { tail -20 $O/${T}_$tag.err; return 1; }
  python3 $R/tools/trace_gaps.py $O/${T}_$tag > $O/${T}_$tag.gaps.txt
  python3 $R/tools/trace_timeline.py $O/${T}_$tag > $O/${T}_$tag.timeline.txt
  echo "== $tag: $(python3 -c "import json;d=json.load(open('$O/${T}_$tag.json'));print(d['value'], d['ms_per_step'])")"
  head -1 $O/${T}_$tag.gaps.txt
  find $O/${T}_$tag -name "*.csv" ! -name "*kernel_trace.csv" -delete
}
run img16 --batch 16 && run img16_graph --batch 16 --graph && \
run vid4 --workload video --batch 4 --clip-len 16 && run vid4_graph --workload video --batch 4 --clip-len 16 --graph && \
run vid1 --workload video --batch 1 --clip-len 16 && run vid1_graph --workload video --batch 1 --clip-len 16 --graph && \
run img1 --batch 1 && run img1_graph --batch 1 --graph
