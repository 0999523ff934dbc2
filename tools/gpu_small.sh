#!/bin/bash
# the reference's own call sizes after a change to the small-grid paths: model / layer / fuzz tests, then image batch 1 / 16,
# video 4 x 16 / 1 x 16 and one dense window (eager, no per-layer events)
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-small}
timeout -k 10 900 python -m pytest tests/test_hip_layers.py tests/test_hip_models.py tests/test_hip_fuzz.py -m gpu -x -q > $O/${T}_tests.log 2>&1 || { tail -40 $O/${T}_tests.log; exit 1; }
tail -1 $O/${T}_tests.log
C="--no-split --no-cpu-baseline --no-train --no-layer-events --steps 100 --warmup 10"
python bench.py --batch 1 $C > $O/${T}_img1.json 2> $O/${T}_1.err || { tail -20 $O/${T}_1.err; exit 1; }
python bench.py --batch 16 $C > $O/${T}_img16.json 2> $O/${T}_2.err || { tail -20 $O/${T}_2.err; exit 1; }
python bench.py --workload video --batch 4 --clip-len 16 $C > $O/${T}_vid4.json 2> $O/${T}_3.err || { tail -20 $O/${T}_3.err; exit 1; }
python bench.py --workload video --batch 1 --clip-len 16 $C > $O/${T}_vid1.json 2> $O/${T}_4.err || { tail -20 $O/${T}_4.err; exit 1; }
python bench.py --workload dense --batch 1 $C > $O/${T}_dense1.json 2> $O/${T}_5.err || { tail -20 $O/${T}_5.err; exit 1; }
python bench.py --no-split --no-cpu-baseline --no-train --no-video --stream-frames 0 > $O/${T}_img512.json 2> $O/${T}_6.err || { tail -20 $O/${T}_6.err; exit 1; }
for f in img1 img16 vid4 vid1 dense1 img512; do python - <<PY
import json
d=json.load(open("$O/${T}_$f.json"))
print("$f", d["value"], d["ms_per_step"])
PY
done
