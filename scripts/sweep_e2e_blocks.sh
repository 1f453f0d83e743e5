#!/bin/bash
# Headline configuration: end-to-end frames/s against the resident grids of k_windows / k_geometry (their LDS is what a
# k_frame block competes with).  Tuning build: scripts/win_variants.sh build tuning "-DYSMR_TUNING"; run through gpurun.
R=$GRAFT_REPO_ROOT
for wb in 0 1024 768; do for gb in 0 768 512; do
  echo -n "window_blocks=$wb geo_blocks=$gb: "
  YSMR_HIP_LIB=$R/scripts/var_tuning.so YSMR_COLLECT_BLOCKS=$wb YSMR_GEO_BLOCKS=$gb python $R/bench.py --cpu-sample 0 2>/dev/null | grep -o '"value": [0-9.]*'
done; done
