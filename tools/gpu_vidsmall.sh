#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-vs}
C="--no-split --no-cpu-baseline --no-train --no-layer-events --steps 100 --warmup 10 --workload video --clip-len 16"
for b in 1 2 3 4; do python bench.py $C --batch $b > $O/${T}_vid$b.json 2> $O/${T}_v$b.err || { tail $O/${T}_v$b.err; exit 1; }; python -c "
import json; d=json.load(open('$O/${T}_vid$b.json')); print('clips $b', d['value'], d['ms_per_step'])"; done
