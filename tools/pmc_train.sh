#!/bin/bash
# MFMA-utilisation counters of the native training step (run on the GPU box): ONE rocprofv3 --pmc pass (kernel-trace only) of
# tools/train_bench.py, summarised per kernel by tools/pmc_mfma_summary.py (normalisation documented there).
# usage: tools/pmc_train.sh <tag> [precision]     -> gpurun_out/pmc_mfma_<tag>.json
set -e
tag=$1; P=${2:-bf16}
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_mfma_$tag
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE \
  --kernel-trace --output-format csv -d $R/gpurun_out/pmc_mfma_$tag -- python3 $R/tools/train_bench.py --clips 32 --steps 2 --warmup 1 --precision $P > /dev/null 2>&1
cd $R
python3 tools/pmc_mfma_summary.py gpurun_out/pmc_mfma_$tag > gpurun_out/pmc_mfma_$tag.json
python3 - <<PY
import json
d = json.load(open("gpurun_out/pmc_mfma_$tag.json"))
for k, v in d["kernels"].items():
    print(k[:78], v["launches"], v["avg_us_profiled"], v.get("mfma_util"), v.get("effective_clock_GHz"))
PY
