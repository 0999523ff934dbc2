#!/bin/bash
# one ConvLSTM step (16x16 map, 128 -> 128) per clip count: cost model's pick (1) / 4-gates-per-wave small-grid kernel (65) /
# 8-wave gate-split kernel (129) / 32x32x2 kernel (9)
set -o pipefail
cd $GRAFT_REPO_ROOT
for n in 1 2 3 4 5 6 8 10 12 16 24 32; do
  for v in 1 65 129 9; do echo -n "n $n variant $v: "; python tools/layer_bench.py lstm --n $n --h 16 --cin 128 --cout 128 --iters 100 --variant $v 2>&1 | tail -1 | awk '{print $6, $7}'; done
done
