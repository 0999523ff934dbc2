#!/bin/bash
# Round-3 GPU session: all GPU tests, then the default bench line (image + video + stream + training objects).
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-r03a}
python -m pytest tests -m gpu -x -q > $O/${T}_tests.log 2>&1 || { tail -40 $O/${T}_tests.log; exit 1; }
tail -3 $O/${T}_tests.log
python bench.py > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err || { tail -20 $O/${T}_bench_default.err; exit 1; }
python - <<PY
import json
d = json.load(open("$O/${T}_bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"])
print({k: round(v["ms"] / v["launches"], 4) for k, v in d["layers"].items()})
for k in ("video", "stream", "training_step", "split_precision"):
    v = d.get(k)
    print(k, json.dumps(v)[:600] if v else None)
PY
