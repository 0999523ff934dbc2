#!/bin/bash
# Winograd mode counters: MFMA utilisation + instruction mix per kernel (image path, --precision winograd).
set -o pipefail
cd $GRAFT_REPO_ROOT
T=${1:-winopmc}
bash tools/pmc_mfma.sh ${T}_image --precision winograd > /dev/null && echo "mfma done"
bash tools/pmc_insts.sh ${T}_image --precision winograd > gpurun_out/pmc_insts_${T}_image.txt && echo "insts done"
python3 - <<PY
import json
d = json.load(open("gpurun_out/pmc_mfma_${T}_image.json"))["kernels"]
for k, e in d.items():
    print(k[:70], e.get("avg_us_profiled"), e.get("mfma_util"), e.get("effective_clock_GHz"), e.get("wave_time_fractions"))
print(open("gpurun_out/pmc_insts_${T}_image.txt").read())
PY
