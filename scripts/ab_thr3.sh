#!/bin/bash
# A/B of builds of the threshold kernel on one box, at the launch shapes that matter: 64 frames alone, 256 frames on the
# grid it takes beside the batch link.  usage: scripts/ab_thr3.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in "$@"; do
    echo -n "$lib b64: "; YSMR_HIP_LIB=$lib python3 scripts/bench_threshold.py --reps 3 --real 2>/dev/null | tail -1
    echo -n "$lib b256 beside: "; YSMR_HIP_LIB=$lib python3 scripts/bench_threshold.py --reps 3 --real --batch 256 --beside 2>/dev/null | tail -1
  done
done
