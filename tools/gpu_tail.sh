#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-r02n}
python -m pytest tests/test_hip_layers.py tests/test_hip_models.py tests/test_hip_fuzz.py tests/test_hip_train_img.py -m gpu -x -q > $O/${T}_tests.log 2>&1 || { tail -40 $O/${T}_tests.log; exit 1; }
tail -2 $O/${T}_tests.log
for i in 1 2; do python tools/layer_bench.py tail --n 128 --h 256 --iters 20; done
python bench.py --no-split --no-train --no-cpu-baseline > $O/${T}_bench_image.json 2> $O/${T}_bi.err || { tail $O/${T}_bi.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/${T}_bench_image.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["layers"]["dec4.3+score"], d["layers"]["dec4.0"])
PY
