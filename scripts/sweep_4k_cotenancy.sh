#!/bin/bash
# BASELINE configs[4] end to end against the resident grids of the detection kernels: who leaves room for the split link
# (tuning build: scripts/build_tuning.sh; run through gpurun).  Per case: frames/s, then the bench line's own device times.
R=$GRAFT_REPO_ROOT
run() { echo -n "$*: "; timeout -k 5 120 env YSMR_HIP_LIB=$R/scripts/var_tuning.so "$@" python3 $R/bench.py --config 4 --cpu-sample 0 --steps 4 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); g = d['diagnostics']
print(round(d['value']), 'frames/s  threshold', round(g['threshold_us_per_batch']['avg']), 'components', round(g['components_us_per_batch']['avg']), 'us/batch  link', round(g['link_us_per_frame']['avg'], 1), 'us/frame (min', round(g['link_us_per_frame']['min'], 1), ')')"; }
run A=0
run YSMR_COLLECT_BLOCKS=1024
run YSMR_COLLECT_BLOCKS=1024 YSMR_GEO_BLOCKS=512
run YSMR_COLLECT_BLOCKS=1024 YSMR_GEO_BLOCKS=512 YSMR_THR_BLOCKS=512
run YSMR_COLLECT_BLOCKS=768 YSMR_GEO_BLOCKS=512 YSMR_THR_BLOCKS=512
run YSMR_COLLECT_BLOCKS=1024 YSMR_GEO_BLOCKS=256 YSMR_THR_BLOCKS=512
run YSMR_THR_BLOCKS=512
run YSMR_GEO_BLOCKS=512
run A=0
