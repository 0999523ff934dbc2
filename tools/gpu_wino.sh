#!/bin/bash
# Winograd prototype session: layer parity tests, then direct vs Winograd per layer at the bench's batch.
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-wino}
timeout -k 10 300 python -m pytest tests/test_hip_layers.py -m gpu -x -q -k "winograd" > $O/${T}_tests.log 2>&1 || { tail -40 $O/${T}_tests.log; exit 1; }
tail -2 $O/${T}_tests.log
timeout -k 10 300 python tools/layer_bench.py winoab --n ${2:-512} --iters 10 2>&1 | tee $O/${T}_ab.txt
