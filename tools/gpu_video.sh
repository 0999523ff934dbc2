#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-r02k}
python -m pytest tests/test_hip_layers.py tests/test_hip_models.py tests/test_hip_fuzz.py -m gpu -x -q > $O/${T}_tests.log 2>&1 || { tail -40 $O/${T}_tests.log; exit 1; }
tail -2 $O/${T}_tests.log
python bench.py --workload video --no-split --no-cpu-baseline > $O/${T}_bench_video.json 2> $O/${T}_bv.err || { tail $O/${T}_bv.err; exit 1; }
python - <<PY
import json
d=json.load(open("$O/${T}_bench_video.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
for k,v in d["layers"].items(): print("  ", k, round(v["ms"]/d["steps"],4))
PY
