#!/bin/bash
# after a kernel change in the bf16-tensor step: layer tests of the training kernels, the two end-to-end gates, timing + trace
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-r03i}
timeout -k 10 900 python -m pytest tests/test_hip_train_bf16.py tests/test_hip_train_ops.py -m gpu -x -q > $O/${T}_tests_ops.log 2>&1 || { tail -40 $O/${T}_tests_ops.log; exit 1; }
tail -1 $O/${T}_tests_ops.log
timeout -k 10 900 python -m pytest tests/test_hip_train_step.py -m gpu -x -q -s -k "loss_curve or bf16" > $O/${T}_tests_step.log 2>&1 || { tail -40 $O/${T}_tests_step.log; exit 1; }
grep -E "loss curve|worst cosine|passed|failed" $O/${T}_tests_step.log
timeout -k 10 300 python tools/train_bench.py --clips 32 --precision bf16 --steps 10 --warmup 3 > $O/${T}_train_bf16.json 2> $O/${T}_train_bf16.err || { tail -20 $O/${T}_train_bf16.err; exit 1; }
cat $O/${T}_train_bf16.json
bash tools/gpu_trace_train.sh $T bf16
