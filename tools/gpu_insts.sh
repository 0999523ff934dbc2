#!/bin/bash
# executed-instruction mix per kernel of a bench command (default: the direct image path)
set -o pipefail
cd $GRAFT_REPO_ROOT
T=${1:-insts}; shift
bash tools/pmc_insts.sh $T "$@" > gpurun_out/pmc_insts_$T.txt && cat gpurun_out/pmc_insts_$T.txt
