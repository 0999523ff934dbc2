#!/bin/bash
# bf16-tensor training mode: layer tests, end-to-end gates, then the step timing of both bf16 forms (32 clips x 10 x 256x256)
# and the kernel-trace stats of the bf16-tensor step.
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-r03h}
timeout -k 10 600 python -m pytest tests/test_hip_train_bf16.py -m gpu -x -q > $O/${T}_tests_bf16.log 2>&1 || { tail -40 $O/${T}_tests_bf16.log; exit 1; }
tail -2 $O/${T}_tests_bf16.log
timeout -k 10 900 python -m pytest tests/test_hip_train_step.py tests/test_hip_train_ops.py -m gpu -x -q -s -k "loss_curve or bf16 or weight_and_data" > $O/${T}_tests_step.log 2>&1 || { tail -40 $O/${T}_tests_step.log; exit 1; }
grep -E "loss curve|worst cosine|passed|failed" $O/${T}_tests_step.log
for P in bf16_operands bf16; do
  timeout -k 10 300 python tools/train_bench.py --clips 32 --precision $P --steps 10 --warmup 3 > $O/${T}_train_$P.json 2> $O/${T}_train_$P.err || { tail -20 $O/${T}_train_$P.err; exit 1; }
  cat $O/${T}_train_$P.json
done
bash tools/gpu_trace_train.sh $T bf16
