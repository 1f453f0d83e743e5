#!/bin/bash
# A/B of builds of the threshold kernel on one box: each library given on the command line, alone on the chip
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in "$@"; do
    echo -n "$lib --real: "; YSMR_HIP_LIB=$lib python3 scripts/bench_threshold.py --reps 3 --real 2>/dev/null | tail -1
  done
done
