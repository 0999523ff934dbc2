#!/bin/bash
# One GPU-box session: tests, then the bench lines (steps joined with && so nothing runs after a failure).
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out
T=${1:-r02a}
python -m pytest tests -m gpu -x -q > $O/${T}_tests.log 2>&1 || { tail -40 $O/${T}_tests.log; exit 1; }
tail -3 $O/${T}_tests.log
python bench.py > $O/${T}_bench_image.json 2> $O/${T}_bench_image.err || { tail -20 $O/${T}_bench_image.err; exit 1; }
python bench.py --workload video > $O/${T}_bench_video.json 2> $O/${T}_bench_video.err || { tail -20 $O/${T}_bench_video.err; exit 1; }
python bench.py --batch 16 --steps 50 --no-split --no-train --no-cpu-baseline > $O/${T}_bench_image_b16.json 2> $O/${T}_b16.err || { tail -20 $O/${T}_b16.err; exit 1; }
python bench.py --workload video --batch 4 --clip-len 16 --steps 50 --no-split --no-cpu-baseline > $O/${T}_bench_video_b4t16.json 2> $O/${T}_v4.err || { tail -20 $O/${T}_v4.err; exit 1; }
python bench.py --workload dense --batch 1 --steps 50 --no-split --no-cpu-baseline > $O/${T}_bench_dense_b1.json 2> $O/${T}_d1.err || { tail -20 $O/${T}_d1.err; exit 1; }
python bench.py --workload dense --batch 64 --steps 20 --no-split --no-cpu-baseline > $O/${T}_bench_dense_b64.json 2> $O/${T}_d64.err || { tail -20 $O/${T}_d64.err; exit 1; }
for f in image video image_b16 video_b4t16 dense_b1 dense_b64; do python - <<PY
import json
d=json.load(open("$O/${T}_bench_$f.json"))
print("$f", d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), (d.get("cpu_baseline") or {}).get("gpu_vs_cpu_max_rel_score_err"))
PY
done
