#!/bin/bash
# HBM traffic of the bench's kernels from PMC counters (run on the GPU box): FETCH_SIZE and WRITE_SIZE in separate
# passes (they do not fit one pass), kernel-trace only.  Writes gpurun_out/pmc_traffic.json.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_$c      # rocprofv3 adds to an existing directory: an earlier workload's launches would be summed in
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-layer-events --no-split --no-train --no-video --no-small --stream-frames 0 "$@" > /dev/null 2>&1
done
cd $R
python3 tools/pmc_traffic_summary.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/pmc_traffic.json
cat gpurun_out/pmc_traffic.json
