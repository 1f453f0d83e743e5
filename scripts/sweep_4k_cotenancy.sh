#!/bin/bash
# BASELINE configs[4] end to end against who leaves room for whom: the link tables in LDS (131 KB, one workgroup) or in
# HBM, and the resident grids of the detection kernels (tuning build: scripts/build_tuning.sh; run through gpurun).
# Per case: frames/s, then the bench line's own per-batch / per-frame device times.
R=$GRAFT_REPO_ROOT
run() { echo -n "$*: "; env YSMR_HIP_LIB=$R/scripts/var_tuning.so "$@" python3 $R/bench.py --config 4 --cpu-sample 0 --steps 4 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); g = d['diagnostics']
print(round(d['value']), 'frames/s  threshold', round(g['threshold_us_per_batch']['avg']), 'components', round(g['components_us_per_batch']['avg']), 'us/batch  link', round(g['link_us_per_frame']['avg'], 1), 'us/frame (min', round(g['link_us_per_frame']['min'], 1), ')')"; }
run A=0
run YSMR_LINK_TABLES=hbm
run YSMR_THR_BLOCKS=512
run YSMR_THR_BLOCKS=256
run YSMR_COLLECT_BLOCKS=1024
run YSMR_COLLECT_BLOCKS=512
run YSMR_GEO_BLOCKS=512
run YSMR_THR_BLOCKS=512 YSMR_COLLECT_BLOCKS=1024 YSMR_GEO_BLOCKS=512
run YSMR_THR_BLOCKS=512 YSMR_COLLECT_BLOCKS=512 YSMR_GEO_BLOCKS=256 YSMR_CLEAR_BLOCKS=256 YSMR_SPARSE_BLOCKS=384
run YSMR_THR_BLOCKS=256 YSMR_COLLECT_BLOCKS=256 YSMR_GEO_BLOCKS=128 YSMR_CLEAR_BLOCKS=128 YSMR_SPARSE_BLOCKS=192
run A=0
