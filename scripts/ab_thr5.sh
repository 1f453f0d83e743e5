#!/bin/bash
# A/B of builds of the threshold kernel at the launch shape the pipeline uses since the end of round 5: 248 frames on the 248
# workgroups it takes beside the batch link, the bench clip.  usage: scripts/ab_thr5.sh lib1.so lib2.so ...
cd $GRAFT_REPO_ROOT
for rep in 1 2 3; do
  for lib in "$@"; do
    echo -n "$lib b248 beside: "; YSMR_HIP_LIB=$lib python3 scripts/bench_threshold.py --reps 5 --real --batch 248 --frames 496 --beside 2>/dev/null | tail -1
  done
done
