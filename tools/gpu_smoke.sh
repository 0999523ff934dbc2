#!/bin/bash
# what the driver runs at round end: build check (no-op when the .so travelled), smoke(), the full -m gpu suite, the default bench line
set -o pipefail
cd $GRAFT_REPO_ROOT
O=gpurun_out; T=${1:-smoke}
python -c "import __graft_entry__ as g; g.smoke()" > $O/${T}_smoke.log 2>&1 || { tail -20 $O/${T}_smoke.log; exit 1; }
tail -1 $O/${T}_smoke.log
timeout -k 10 1000 python -m pytest tests -m gpu -q > $O/${T}_tests.log 2>&1 || { tail -40 $O/${T}_tests.log; exit 1; }
tail -1 $O/${T}_tests.log
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/${T}_bench.json 2> $O/${T}_bench.err || { tail -20 $O/${T}_bench.err; exit 1; }
python -c "
import json; d=json.load(open('$O/${T}_bench.json')); print(d['metric'], d['value'], d['unit'], d['ms_per_step'], d['roofline']['frac'], d['cpu_baseline']['value'], d['training_step']['bf16_precision']['value'])"
