#!/bin/bash
# full GPU suite, the default bench line, MFMA counters of the bf16 training step
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-r03n}
timeout -k 10 1000 python -m pytest tests -m gpu -x -q > $O/${T}_tests.log 2>&1 || { tail -40 $O/${T}_tests.log; exit 1; }
tail -2 $O/${T}_tests.log
timeout -k 10 600 python bench.py > $O/${T}_bench_default.json 2> $O/${T}_bench_default.err || { tail -20 $O/${T}_bench_default.err; exit 1; }
python - <<PY
import json
d = json.load(open("$O/${T}_bench_default.json"))
print(d["value"], d["ms_per_step"], d["roofline"]["frac"], "split", d["split_precision"]["value"])
ts = d["training_step"]
for k in ("fp32", "split_precision", "bf16_precision"):
    r = ts if k == "fp32" else ts[k]
    rf = r["roofline"]
    print(k, r["value"], r["ms_per_step"], "mfma", rf["achieved"], rf["frac"], rf["ms"], "bn", rf["batchnorm_passes"])
    if k == "bf16_precision":
        for n, g in rf["groups"].items(): print("   ", n, g)
PY
bash tools/pmc_train.sh ${T}_bf16 bf16 > $O/${T}_pmc_train.log 2>&1 || { tail -20 $O/${T}_pmc_train.log; exit 1; }
tail -5 $O/${T}_pmc_train.log
