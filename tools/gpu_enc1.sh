#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-r02h}
python -m pytest tests/test_hip_layers.py tests/test_hip_models.py tests/test_hip_fuzz.py -m gpu -x -q > $O/${T}_tests.log 2>&1 || { tail -40 $O/${T}_tests.log; exit 1; }
tail -2 $O/${T}_tests.log
for v in 1 17 1 17; do python tools/layer_bench.py c3fused --n 128 --h 256 --iters 20 --variant $v; done
python bench.py --no-split --no-train --no-cpu-baseline > $O/${T}_bench_image.json 2> $O/${T}_bi.err || { tail $O/${T}_bi.err; exit 1; }
python bench.py --no-split --no-train --no-cpu-baseline --conv-variant 17 > $O/${T}_bench_image_c3mfma.json 2> $O/${T}_bi2.err || { tail $O/${T}_bi2.err; exit 1; }
for f in bench_image bench_image_c3mfma; do python - <<PY
import json
d=json.load(open("$O/${T}_$f.json"))
print("$f", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["layers"]["enc1.0+enc1.3+pool"])
PY
done
