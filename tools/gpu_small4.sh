#!/bin/bash
# The reference's call sizes with per-layer events: where one batch-16 image call and one 16-frame window spend their time.
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-small4}
python bench.py --batch 16 --steps 300 --warmup 20 --no-split --no-train --no-cpu-baseline --no-video --no-small --stream-frames 0 > $O/${T}_img16.json 2>/dev/null
python bench.py --batch 16 --steps 300 --warmup 20 --no-split --no-train --no-cpu-baseline --no-video --no-small --stream-frames 0 --no-layer-events > $O/${T}_img16_noev.json 2>/dev/null
python bench.py --workload video --batch 1 --clip-len 16 --steps 300 --warmup 20 --no-split --no-train --no-cpu-baseline --no-small > $O/${T}_vid1.json 2>/dev/null
python bench.py --workload video --batch 1 --clip-len 16 --steps 300 --warmup 20 --no-split --no-train --no-cpu-baseline --no-small --no-layer-events > $O/${T}_vid1_noev.json 2>/dev/null
python - <<PY
import json
for n in ("img16", "vid1"):
    d = json.load(open("$O/${T}_%s.json" % n)); e = json.load(open("$O/${T}_%s_noev.json" % n))
    print(n, "with events", d["ms_per_step"], "without", e["ms_per_step"])
    print({k: round(v["ms"] / v["launches"] * 1e3, 1) for k, v in d["layers"].items()}, "us per launch; sum", round(sum(v["ms"] for v in d["layers"].values()) / d["steps"], 4))
PY
