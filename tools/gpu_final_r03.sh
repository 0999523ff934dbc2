#!/bin/bash
# round-3 closing evidence: full -m gpu suite, the default bench line, PMC traffic of the image and video bench commands
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-r03final}
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/${T}_tests.log 2>&1 || { tail -40 $O/${T}_tests.log; exit 1; }
tail -2 $O/${T}_tests.log
timeout -k 10 600 python bench.py > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err || { tail -20 $O/${T}_bench_default.err; exit 1; }
python - <<PY
import json
d = json.load(open("$O/${T}_bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"]["whole_path_tflops"], "split", d["split_precision"]["value"], "cpu", d["cpu_baseline"]["value"], d["cpu_baseline"].get("gpu_vs_cpu_max_rel_score_err"))
print({k: round(v["ms"] / v["launches"], 4) for k, v in d["layers"].items()})
v = d["video"]; print("video", v["value"], v["ms_per_step"], v["roofline"]["frac"]); print("stream", d["stream"]["seconds"], d["stream"]["value"])
ts = d["training_step"]
for k in ("fp32", "split_precision", "bf16_precision"):
    r = ts if k == "fp32" else ts[k]
    rf = r["roofline"]
    print(k, r["value"], r["ms_per_step"], "mfma", rf["achieved"], rf["frac"], rf["ms"], "bn", rf["batchnorm_passes"])
    if k == "bf16_precision":
        for n, g in rf["groups"].items(): print("   ", n, g)
PY
bash tools/pmc_traffic.sh > /dev/null 2>&1 && cp gpurun_out/pmc_traffic.json gpurun_out/${T}_pmc_traffic_image.json && echo "traffic image done"
bash tools/pmc_traffic.sh --workload video > /dev/null 2>&1 && cp gpurun_out/pmc_traffic.json gpurun_out/${T}_pmc_traffic_video.json && echo "traffic video done"
