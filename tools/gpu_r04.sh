#!/bin/bash
# Round-4 GPU session: the GPU tests from a given file on (default: all), then the default bench line.
#   gpurun -- 'bash tools/gpu_r04.sh <tag> [pytest args...]'
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-r04}; shift
ARGS=${@:-tests}
python -m pytest $ARGS -m gpu -x -q > $O/${T}_tests.log 2>&1 || { tail -60 $O/${T}_tests.log; exit 1; }
tail -3 $O/${T}_tests.log
python bench.py > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err || { tail -20 $O/${T}_bench_default.err; exit 1; }
python - <<PY
import json
d = json.load(open("$O/${T}_bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["traffic_source"])
print({k: round(v["ms"] / v["launches"], 4) for k, v in d["layers"].items()})
for k in ("video", "stream", "training_step", "split_precision", "reference_call_sizes"):
    v = d.get(k)
    print(k, json.dumps(v)[:700] if v else None)
PY
