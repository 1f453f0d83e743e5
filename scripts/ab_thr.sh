#!/bin/bash
# A/B of two builds of the threshold kernel on one box: scripts/var_thr_old.so against the in-tree library
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
  for lib in scripts/var_thr_old.so ysmr_amd/csrc/libysmr_hip.so; do
    for mode in "" "--real"; do
      echo -n "$lib $mode: "; YSMR_HIP_LIB=$lib python3 scripts/bench_threshold.py --reps 3 $mode 2>/dev/null | tail -1
    done
  done
done
