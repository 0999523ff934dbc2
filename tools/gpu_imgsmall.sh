#!/bin/bash
# the image path at the reference's call sizes (one image, 16 images), with and without the gate-split kernel on plain layers
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests/test_hip_layers.py tests/test_hip_models.py tests/test_hip_fuzz.py -m gpu -x -q 2>&1 | tail -3 || exit 1
C="--no-split --no-cpu-baseline --no-train --no-layer-events --steps 200 --warmup 20"
for b in 1 2 4 8 16 32; do for v in 65 1; do echo -n "batch $b variant $v: "; python bench.py $C --batch $b --conv-variant $v 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'])"; done; done
