#!/bin/bash
# one 16-frame window / up to 4 clips x 16: the reference's own video call sizes
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-vs}
timeout -k 10 600 python -m pytest tests/test_hip_layers.py tests/test_hip_models.py -m gpu -x -q -k "lstm or video or window or small" 2>&1 | tail -3 || exit 1
bash tools/gpu_lstm_slope.sh 2>&1 | tee $O/${T}_slope.txt
C="--no-split --no-cpu-baseline --no-train --no-layer-events --steps 100 --warmup 10 --workload video --clip-len 16"
for b in 1 2 4 8; do python bench.py $C --batch $b > $O/${T}_vid$b.json 2> $O/${T}_v$b.err || { tail $O/${T}_v$b.err; exit 1; }; python -c "
import json; d=json.load(open('$O/${T}_vid$b.json')); print('clips $b', d['value'], d['ms_per_step'])"; done
