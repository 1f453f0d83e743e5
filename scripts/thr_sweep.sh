#!/bin/bash
# Threshold kernel alone: tuning builds (scripts/var_*.so, EXTRA=-DYSMR_TUNING) x segment heights x resident grids.
R=$GRAFT_REPO_ROOT
b() { python $R/scripts/bench_threshold.py "$@" | tail -1; }
echo -n "shipped library:            "; b
for v in tune; do
  for sh in 45 58 116; do
    for blocks in 768 1024; do
      echo -n "$v seg_h=$sh blocks=$blocks: "; YSMR_HIP_LIB=$R/scripts/var_$v.so YSMR_SEG_H=$sh YSMR_THR_BLOCKS=$blocks b
    done
  done
done
echo -n "tune seg_h=116 blocks=768 real frames: "; YSMR_HIP_LIB=$R/scripts/var_tune.so YSMR_SEG_H=116 b --real
