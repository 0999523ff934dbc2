#!/bin/bash
# SQ counters of the training-step kernels (run on the GPU box): two --pmc passes (kernel-trace only), summary per kernel.
set -e
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/pmc_train_a -- python3 $R/tools/train_bench.py --clips 32 --steps 1 --warmup 1 > /dev/null 2>&1
rocprofv3 --pmc SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_WAVES SQ_ACTIVE_INST_SCA SQ_INSTS_SALU SQ_INSTS_VMEM_RD --kernel-trace --output-format csv -d $R/gpurun_out/pmc_train_b -- python3 $R/tools/train_bench.py --clips 32 --steps 1 --warmup 1 > /dev/null 2>&1
cd $R
python3 tools/pmc_summary.py gpurun_out/pmc_train_a gpurun_out/pmc_train_b
