#!/bin/bash
# After a change to the stand-alone first-layer kernel: every test that runs it (scoring, training forward, bf16 output), then the
# video / winograd numbers and the training step.
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-c3}
timeout -k 10 900 python -m pytest tests/test_hip_layers.py tests/test_hip_models.py tests/test_hip_fuzz.py tests/test_hip_train_ops.py tests/test_hip_train_bf16.py tests/test_hip_train_step.py -m gpu -x -q > $O/${T}_tests.log 2>&1 || { tail -40 $O/${T}_tests.log; exit 1; }
tail -2 $O/${T}_tests.log
timeout -k 10 600 python bench.py --no-cpu-baseline --no-small --stream-frames 0 > $O/${T}_bench.json 2> $O/${T}_bench.err || { tail -20 $O/${T}_bench.err; exit 1; }
python - <<PY
import json
d = json.load(open("$O/${T}_bench.json"))
print("direct", d["value"], d["ms_per_step"], d["roofline"]["frac"])
w = d["winograd_precision"]; print("winograd", w["value"], w["ms_per_step"], {k: v["ms"] for k, v in w["layers"].items()})
v = d["video"]
print("video direct", v["value"], v["ms_per_step"], {k: round(x["ms"] / x["launches"], 4) for k, x in v["layers"].items()})
print("video winograd", v["winograd_precision"]["value"], v["winograd_precision"]["ms_per_step"], {k: x["ms"] for k, x in v["winograd_precision"]["layers"].items()})
t = d["training_step"]
for k in ("", "split_precision", "bf16_precision", "winograd_precision"):
    r = t[k] if k else t
    print("train", k or "fp32", r["value"], r["ms_per_step"], {g: x["ms"] for g, x in r["roofline"]["groups"].items()})
PY
