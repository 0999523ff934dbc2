#!/bin/bash
# Winograd mode end to end: model-level parity tests, then the image and video bench lines with their winograd objects.
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-wino2}
timeout -k 10 600 python -m pytest tests/test_hip_layers.py tests/test_hip_models.py -m gpu -x -q -k "winograd or split_precision or blob or c_abi" > $O/${T}_tests.log 2>&1 || { tail -40 $O/${T}_tests.log; exit 1; }
tail -2 $O/${T}_tests.log
timeout -k 10 600 python bench.py --no-train --no-cpu-baseline --no-small --stream-frames 0 > $O/${T}_bench.json 2> $O/${T}_bench.err || { tail -20 $O/${T}_bench.err; exit 1; }
python - <<PY
import json
d = json.load(open("$O/${T}_bench.json"))
print("direct", d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k in ("split_precision", "winograd_precision"):
    print(k, json.dumps(d[k])[:1500])
v = d["video"]
print("video direct", v["value"], v["ms_per_step"]); print("video winograd", json.dumps(v["winograd_precision"])[:900])
PY
