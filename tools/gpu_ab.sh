#!/bin/bash
# quick A/B: layer + model tests, then the image bench with per-layer numbers
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-ab}
python -m pytest tests/test_hip_layers.py tests/test_hip_models.py tests/test_hip_fuzz.py -m gpu -x -q > $O/${T}_tests.log 2>&1 || { tail -40 $O/${T}_tests.log; exit 1; }
tail -2 $O/${T}_tests.log
for i in 1 2; do python bench.py --no-split --no-train --no-cpu-baseline > $O/${T}_bench_image_$i.json 2> $O/${T}_bi.err || { tail $O/${T}_bi.err; exit 1; }; done
python - <<PY
import json
for i in (1,2):
    d=json.load(open("$O/${T}_bench_image_%d.json"%i))
    print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
    print("   ", {k: (round(v["ms"]/d["steps"],3), v.get("tflops")) for k,v in d["layers"].items() if k in ("enc1.0+enc1.3+pool","dec3.3","enc2.0","dec2.3")})
PY
