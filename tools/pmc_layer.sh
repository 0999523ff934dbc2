#!/bin/bash
# usage: tools/pmc_layer.sh <tag> <layer_bench args...>   (run on the GPU box; writes gpurun_out/pmc_<tag>_*.csv)
set -e
tag=$1; shift
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_a -- python3 $R/tools/layer_bench.py "$@" --iters 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES --kernel-trace --output-format csv -d $R/gpurun_out/pmc_${tag}_b -- python3 $R/tools/layer_bench.py "$@" --iters 3 > /dev/null 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_${tag}_a gpurun_out/pmc_${tag}_b
