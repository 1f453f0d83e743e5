#!/bin/bash
# k_clear's resident grid (YSMR_CLEAR_BLOCKS; needs a -DYSMR_TUNING build: scripts/build_tuning.sh) against the labelling chain's
# time in the detection-only configuration and in the headline one
cd $GRAFT_REPO_ROOT
for c in 1 2; do
for b in 512 1024 2048; do
  YSMR_HIP_LIB=scripts/var_tuning.so YSMR_CLEAR_BLOCKS=$b python3 bench.py --config $c --cpu-sample 0 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline()); g = d['diagnostics']
print('config $c clear blocks $b:', round(d['value']), 'frames/s  components us/batch', round(g['components_us_per_batch']['avg'], 1))"
done
done
