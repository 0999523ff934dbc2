#!/bin/bash
# small-grid ConvLSTM step: time against the number of 32-channel chunks of its K loop (fixed cost vs cost per chunk)
set -o pipefail
cd $GRAFT_REPO_ROOT
for cin in 32 64 128 256 384; do python tools/layer_bench.py lstm --n 1 --h 16 --cin $cin --cout 128 --iters 200 2>&1 | tail -1; done
for n in 2 4 8; do python tools/layer_bench.py lstm --n $n --h 16 --cin 128 --cout 128 --iters 200 2>&1 | tail -1; done
