#!/bin/bash
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-r02b}
python -m pytest tests -m gpu -q > $O/${T}_tests.log 2>&1 || { tail -40 $O/${T}_tests.log; exit 1; }
tail -2 $O/${T}_tests.log
python bench.py --workload video --batch 4 --clip-len 16 --steps 50 --no-split --no-cpu-baseline > $O/${T}_bench_video_b4t16.json 2> $O/${T}_v4.err || { tail -20 $O/${T}_v4.err; exit 1; }
python bench.py --workload video --batch 4 --clip-len 16 --steps 50 --no-split --no-cpu-baseline --no-wavefront > $O/${T}_bench_video_b4t16_nowf.json 2> $O/${T}_v4n.err || { tail -20 $O/${T}_v4n.err; exit 1; }
python bench.py --workload video --batch 4 --clip-len 16 --steps 50 --no-split --no-cpu-baseline --no-layer-events > $O/${T}_bench_video_b4t16_noev.json 2> $O/${T}_v4e.err || { tail -20 $O/${T}_v4e.err; exit 1; }
python bench.py --workload dense --batch 1 --steps 50 --no-split --no-cpu-baseline --no-layer-events > $O/${T}_bench_dense_b1_noev.json 2> $O/${T}_d1e.err || { tail -20 $O/${T}_d1e.err; exit 1; }
python bench.py --workload dense --batch 1 --steps 50 --no-split --no-cpu-baseline > $O/${T}_bench_dense_b1.json 2> $O/${T}_d1.err || { tail -20 $O/${T}_d1.err; exit 1; }
python bench.py --workload video --batch 16 --steps 30 --no-split --no-cpu-baseline > $O/${T}_bench_video_b16.json 2> $O/${T}_v16.err || { tail -20 $O/${T}_v16.err; exit 1; }
python bench.py --workload video --no-split --no-cpu-baseline > $O/${T}_bench_video.json 2> $O/${T}_v.err || { tail -20 $O/${T}_v.err; exit 1; }
for f in video_b4t16 video_b4t16_nowf video_b4t16_noev dense_b1 dense_b1_noev video_b16 video; do python - <<PY
import json
d=json.load(open("$O/${T}_bench_$f.json"))
l=d.get("layers") or {}
print("$f", d["value"], d["ms_per_step"], "lstm ms/step", round(l["convlstm"]["ms"]/d["steps"],3) if l else None)
PY
done
