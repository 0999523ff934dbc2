#!/bin/bash
# Split-fp16 scoring path profiled: kernel-trace stats, MFMA-utilisation counters and instruction mix
# of `bench.py --precision split` (image) in separate rocprofv3 passes.
set -o pipefail
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out; T=${1:-r03s}
cd $R
python bench.py --precision split --no-train --no-cpu-baseline --no-split > $O/${T}_bench_split.json 2> $O/${T}_bench_split.err || { tail -20 $O/${T}_bench_split.err; exit 1; }
python - <<PY
import json
d = json.load(open("$O/${T}_bench_split.json"))
print("split image", d["value"], d["ms_per_step"])
print({k: round(v["ms"] / v["launches"], 4) for k, v in d["layers"].items()})
PY
bash tools/gpu_trace.sh ${T}_split --precision split --steps 5 --warmup 2 --no-train --no-cpu-baseline --no-split --no-layer-events && \
bash tools/pmc_mfma.sh ${T}_split --precision split > $O/${T}_pmc_mfma.log 2>&1 && \
bash tools/pmc_insts.sh ${T}_split --precision split > $O/${T}_pmc_insts.txt 2>&1 && cat $O/${T}_pmc_insts.txt
