#!/bin/bash
# Short GPU check after a kernel change: layer / model / fuzz tests, then the image and video bench lines (no CPU leg).
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-q}
python -m pytest tests/test_hip_layers.py tests/test_hip_models.py tests/test_hip_fuzz.py -m gpu -x -q > $O/${T}_tests.log 2>&1 || { tail -40 $O/${T}_tests.log; exit 1; }
tail -1 $O/${T}_tests.log
python bench.py --no-split --no-train --no-cpu-baseline > $O/${T}_bench_image.json 2> $O/${T}_bi.err || { tail $O/${T}_bi.err; exit 1; }
python bench.py --workload video --no-split --no-train --no-cpu-baseline > $O/${T}_bench_video.json 2> $O/${T}_bv.err || { tail $O/${T}_bv.err; exit 1; }
python - <<PY
import json
for w in ("image", "video"):
    d = json.load(open("$O/${T}_bench_%s.json" % w))
    print(w, d["value"], d["ms_per_step"], d["roofline"]["frac"])
    print("   ", {k: round(v["ms"] / v["launches"], 4) for k, v in d["layers"].items()})
PY
