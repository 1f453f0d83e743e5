#!/bin/bash
# A/B of two builds of the library in ONE gpurun call (boxes differ by a percent or two): scripts/var_old.so against
# the library in the tree, end-to-end frames/s, alternating.
R=$GRAFT_REPO_ROOT
for i in 1 2 3 4; do
  echo -n "tree: "; python $R/bench.py --cpu-sample 0 $BENCH_ARGS 2>/dev/null | grep -o '"value": [0-9.]*'
  echo -n "old:  "; YSMR_HIP_LIB=$R/scripts/var_old.so python $R/bench.py --cpu-sample 0 $BENCH_ARGS 2>/dev/null | grep -o '"value": [0-9.]*'
done
