#!/bin/bash
# k_windows' resident grid (YSMR_COLLECT_BLOCKS; a -DYSMR_TUNING build: scripts/build_tuning.sh) against the labelling chain's time,
# detection alone and in the pipeline (66 registers: seven workgroups per compute unit = 1792 resident, 2048 are launched)
cd $GRAFT_REPO_ROOT
for rep in 1 2; do
for b in ${BLOCKS:-1536 1792 2048 3584}; do
for c in 1 2; do
  YSMR_HIP_LIB=scripts/var_tuning.so YSMR_COLLECT_BLOCKS=$b python3 bench.py --config $c --cpu-sample 0 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline()); g = d['diagnostics']
print('config $c window blocks $b:', round(d['value']), 'frames/s  components us/batch', round(g['components_us_per_batch']['avg'], 1))"
done
done
done
