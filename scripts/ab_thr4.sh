#!/bin/bash
# by-deletion / A-B of builds of the threshold kernel on BACKGROUND NOISE (no blobs: every tile takes the calm path, nothing is
# listed -- so that a build with a part deleted differs from the complete one by that part only), 256 frames beside the batch link
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in "$@"; do
    echo -n "$lib noise b256 beside: "; YSMR_HIP_LIB=$lib python3 scripts/bench_threshold.py --reps 3 --batch 256 --beside 2>/dev/null | tail -1
  done
done
