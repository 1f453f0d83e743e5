#!/bin/bash
R=$GRAFT_REPO_ROOT
for v in "" lds lds_pf2 pf2 pf4; do
  for sh in 45 56; do
    if [ -z "$v" ]; then echo -n "default seg_h=$sh: "; YSMR_SEG_H=$sh python $R/scripts/bench_threshold.py | tail -1
    else echo -n "$v seg_h=$sh: "; YSMR_SEG_H=$sh YSMR_HIP_LIB=$R/scripts/var_$v.so python $R/scripts/bench_threshold.py | tail -1; fi
  done
done
YSMR_HIP_LIB=$R/scripts/var_lds.so python -m pytest $R/tests/test_gpu_detect.py -m gpu -q 2>&1 | tail -1
