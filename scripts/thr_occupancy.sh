#!/bin/bash
# Threshold kernel: same work (6144 items of 58 rows), 1 / 2 / 3 resident waves per SIMD, plus 4 with 8192 items of 44 rows.
R=$GRAFT_REPO_ROOT
b() { python $R/scripts/bench_threshold.py "$@" | tail -1; }
for blocks in 256 512 768; do echo -n "seg_h=58 blocks=$blocks: "; YSMR_HIP_LIB=$R/scripts/var_tune.so YSMR_SEG_H=58 YSMR_THR_BLOCKS=$blocks b; done
for blocks in 256 512 1024; do echo -n "seg_h=44 blocks=$blocks: "; YSMR_HIP_LIB=$R/scripts/var_tune.so YSMR_SEG_H=44 YSMR_THR_BLOCKS=$blocks b; done
