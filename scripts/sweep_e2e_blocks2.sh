#!/bin/bash
# End-to-end and detection-only frames/s for a few choices of the resident grids of k_windows / k_geometry (tuning build
# of the library: scripts/win_variants.sh build tuning "-DYSMR_TUNING"; run through gpurun).
R=$GRAFT_REPO_ROOT
for cfg in "0 0" "2048 0" "2048 1536" "1024 1536" "1536 768"; do set -- $cfg
  echo -n "window_blocks=$1 geo_blocks=$2: e2e "
  YSMR_HIP_LIB=$R/scripts/var_tuning.so YSMR_COLLECT_BLOCKS=$1 YSMR_GEO_BLOCKS=$2 python $R/bench.py --cpu-sample 0 2>/dev/null | grep -o '"value": [0-9.]*' | tr '\n' ' '
  echo -n " detect-only "
  YSMR_HIP_LIB=$R/scripts/var_tuning.so YSMR_COLLECT_BLOCKS=$1 YSMR_GEO_BLOCKS=$2 python $R/bench.py --cpu-sample 0 --config 1 2>/dev/null | grep -o '"value": [0-9.]*'
done
