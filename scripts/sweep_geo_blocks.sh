#!/bin/bash
# k_geometry's grid (YSMR_GEO_BLOCKS; a -DYSMR_TUNING build) against the labelling chain's time, detection alone and in the pipeline
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for b in ${BLOCKS:-1536 2048 3072 4096}; do
for c in 1 2; do
  YSMR_HIP_LIB=scripts/var_tuning.so YSMR_GEO_BLOCKS=$b python3 bench.py --config $c --cpu-sample 0 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline()); g = d['diagnostics']
print('config $c geometry blocks $b:', round(d['value']), 'frames/s  components us/batch', round(g['components_us_per_batch']['avg'], 1))"
done
done
done
