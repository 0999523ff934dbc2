#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-r02g}
python -m pytest tests/test_hip_train_step.py tests/test_hip_train_ops.py tests/test_hip_layers.py -m gpu -x -q -s -k "loss_curve or bf16 or gradients or convt2x2 or conv3x3" > $O/${T}_tests.log 2>&1 || { tail -60 $O/${T}_tests.log; exit 1; }
grep -n "loss curve\|bf16 vs\|passed\|failed" $O/${T}_tests.log | tail
for p in fp32 split bf16; do python tools/train_bench.py --clips 32 --precision $p > $O/${T}_train_$p.json 2> $O/${T}_train_$p.err || { tail -20 $O/${T}_train_$p.err; exit 1; }; cat $O/${T}_train_$p.json; done
