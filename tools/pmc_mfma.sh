#!/bin/bash
# MFMA-utilisation counters of the bench's kernels (run on the GPU box): ONE rocprofv3 pass with SQ + GRBM counters
# (--pmc only ever together with --kernel-trace), summarised per kernel by tools/pmc_mfma_summary.py.
# usage: tools/pmc_mfma.sh <tag> [bench.py args...]    -> gpurun_out/pmc_mfma_<tag>.json
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_mfma_$tag
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-layer-events --no-split --no-train --no-video --no-small --stream-frames 0 "$@" > /dev/null 2>&1
cd $R
python3 tools/pmc_mfma_summary.py gpurun_out/pmc_mfma_$tag > gpurun_out/pmc_mfma_$tag.json
cat gpurun_out/pmc_mfma_$tag.json
