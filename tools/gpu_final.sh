#!/bin/bash
# Round evidence: bench lines (image with the 100k-frame stream, video, small batches, dense, uint8 ingest) for profiles/.
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-r02}
python bench.py --stream-frames 100000 > $O/${T}_bench_image.json 2> $O/${T}_e1.err || { tail $O/${T}_e1.err; exit 1; }
python bench.py --workload video > $O/${T}_bench_video.json 2> $O/${T}_e2.err || { tail $O/${T}_e2.err; exit 1; }
python bench.py --ingest u8 --no-split --no-train > $O/${T}_bench_image_u8_ingest.json 2> $O/${T}_e3.err || { tail $O/${T}_e3.err; exit 1; }
S="--no-split --no-train --no-cpu-baseline --no-layer-events --steps 100 --warmup 10"
python bench.py --batch 16 $S > $O/${T}_bench_image_b16.json 2> $O/${T}_e4.err || { tail $O/${T}_e4.err; exit 1; }
python bench.py --batch 1 $S > $O/${T}_bench_image_b1.json 2> $O/${T}_e5.err || { tail $O/${T}_e5.err; exit 1; }
python bench.py --workload video --batch 4 --clip-len 16 $S > $O/${T}_bench_video_b4t16.json 2> $O/${T}_e6.err || { tail $O/${T}_e6.err; exit 1; }
python bench.py --workload video --batch 1 --clip-len 16 $S > $O/${T}_bench_video_b1t16.json 2> $O/${T}_e7.err || { tail $O/${T}_e7.err; exit 1; }
python bench.py --workload dense --batch 1 $S > $O/${T}_bench_dense_b1.json 2> $O/${T}_e8.err || { tail $O/${T}_e8.err; exit 1; }
python bench.py --workload dense --batch 64 --no-split --no-cpu-baseline > $O/${T}_bench_dense_windows.json 2> $O/${T}_e9.err || { tail $O/${T}_e9.err; exit 1; }
for p in fp32 split bf16; do python tools/train_bench.py --clips 32 --precision $p > $O/${T}_train_$p.json 2> $O/${T}_t$p.err || { tail $O/${T}_t$p.err; exit 1; }; done
for f in bench_image bench_video bench_image_u8_ingest bench_image_b16 bench_image_b1 bench_video_b4t16 bench_video_b1t16 bench_dense_b1 bench_dense_windows; do python - <<PY
import json
d=json.load(open("$O/${T}_$f.json"))
print("$f", d["value"], d["ms_per_step"], (d.get("roofline") or {}).get("frac"), (d.get("stream") or {}).get("value"))
PY
done
cat $O/${T}_train_*.json | cut -c1-200
