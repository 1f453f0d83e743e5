#!/bin/bash
# BASELINE configs[4] end to end against who leaves room for whom: the link tables in LDS (131 KB, one workgroup) or in
# HBM, and the resident grids of the detection kernels (tuning build: scripts/build_tuning.sh; run through gpurun).
R=$GRAFT_REPO_ROOT
run() { echo -n "$*: "; env YSMR_HIP_LIB=$R/scripts/var_tuning.so "$@" python3 $R/bench.py --config 4 --cpu-sample 0 --steps 4 2>/dev/null | grep -o '"value": [0-9.]*' ; }
run A=0
run YSMR_LINK_TABLES=hbm
run YSMR_THR_BLOCKS=512
run YSMR_COLLECT_BLOCKS=1024
run YSMR_COLLECT_BLOCKS=512
run YSMR_GEO_BLOCKS=512
run YSMR_GEO_BLOCKS=256
run YSMR_THR_BLOCKS=512 YSMR_COLLECT_BLOCKS=1024 YSMR_GEO_BLOCKS=512
run YSMR_THR_BLOCKS=512 YSMR_COLLECT_BLOCKS=1024 YSMR_GEO_BLOCKS=512 YSMR_LINK_TABLES=hbm
run YSMR_THR_BLOCKS=512 YSMR_COLLECT_BLOCKS=512 YSMR_GEO_BLOCKS=256 YSMR_CLEAR_BLOCKS=256 YSMR_SPARSE_BLOCKS=384
run YSMR_THR_BLOCKS=512 YSMR_COLLECT_BLOCKS=512 YSMR_GEO_BLOCKS=256 YSMR_CLEAR_BLOCKS=256 YSMR_SPARSE_BLOCKS=384 YSMR_LINK_TABLES=hbm
run A=0
