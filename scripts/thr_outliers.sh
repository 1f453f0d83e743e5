#!/bin/bash
# how often a threshold launch of the pipeline takes twice its time: N plain bench runs per library, alternating (scripts/thr_outliers.sh N lib1 lib2 ...)
cd $GRAFT_REPO_ROOT
N=$1; shift
for rep in $(seq 1 $N); do
for lib in "$@"; do
  unset YSMR_HIP_LIB; [ "$lib" = default ] || export YSMR_HIP_LIB=$lib
  python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline()); g = d['diagnostics']; l = g['threshold_us_by_launch']; m = sorted(l)[len(l) // 2]
print('$lib', round(d['value']), 'frames/s  frac', round(d['roofline']['frac'], 4), ' median', m, 'us  launches over 1.3 x:', [(i, x) for i, x in enumerate(l) if x > 1.3 * m])"
done
done
