#!/bin/bash
# A/B/C of library builds (scripts/var_<name>.so) in ONE gpurun call, alternating: boxes differ by a percent or two.
#   VARIANTS="base grid claimrow" BENCH_ARGS="--config 4 --steps 4" bash scripts/ab_variants.sh
R=$GRAFT_REPO_ROOT
for i in 1 2 3 4; do
  for v in $VARIANTS; do
    echo -n "$v: "; YSMR_HIP_LIB=$R/scripts/var_$v.so timeout -k 5 120 python3 $R/bench.py --cpu-sample 0 $BENCH_ARGS 2>/dev/null | grep -o '"value": [0-9.]*'
  done
done
