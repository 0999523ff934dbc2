#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-r02f}
python -m pytest tests/test_hip_models.py -m gpu -x -q -k "graph or thread or golden" > $O/${T}_tests.log 2>&1 || { tail -40 $O/${T}_tests.log; exit 1; }
tail -2 $O/${T}_tests.log
C="--no-split --no-cpu-baseline --no-train --no-layer-events --steps 100 --warmup 10"
python bench.py --batch 16 $C > $O/${T}_img16.json 2> $O/${T}_1.err || { tail -20 $O/${T}_1.err; exit 1; }
python bench.py --batch 16 $C --graph > $O/${T}_img16_graph.json 2> $O/${T}_2.err || { tail -20 $O/${T}_2.err; exit 1; }
python bench.py --workload video --batch 4 --clip-len 16 $C > $O/${T}_vid4.json 2> $O/${T}_3.err || { tail -20 $O/${T}_3.err; exit 1; }
python bench.py --workload video --batch 4 --clip-len 16 $C --graph > $O/${T}_vid4_graph.json 2> $O/${T}_4.err || { tail -20 $O/${T}_4.err; exit 1; }
python bench.py --workload video --batch 1 --clip-len 16 $C > $O/${T}_vid1.json 2> $O/${T}_5.err || { tail -20 $O/${T}_5.err; exit 1; }
python bench.py --workload video --batch 1 --clip-len 16 $C --graph > $O/${T}_vid1_graph.json 2> $O/${T}_6.err || { tail -20 $O/${T}_6.err; exit 1; }
python bench.py --batch 1 $C > $O/${T}_img1.json 2> $O/${T}_7.err || { tail -20 $O/${T}_7.err; exit 1; }
python bench.py --batch 1 $C --graph > $O/${T}_img1_graph.json 2> $O/${T}_8.err || { tail -20 $O/${T}_8.err; exit 1; }
for f in img16 img16_graph vid4 vid4_graph vid1 vid1_graph img1 img1_graph; do python - <<PY
import json
d=json.load(open("$O/${T}_$f.json"))
print("$f", d["value"], d["ms_per_step"])
PY
done
