#!/bin/bash
# detection alone (bench.py --config 1) with k_geometry at 8 / 4 / 2 lanes per component and several resident grids
cd $GRAFT_REPO_ROOT
pick='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); d=r["diagnostics"]; print("%.1f k frames/s, threshold %.1f us, components %.1f us" % (r["value"]/1e3, d["threshold_us_per_batch"]["avg"], d["components_us_per_batch"]["avg"]))'
for lib in scripts/var_tuning.so scripts/var_geo4.so scripts/var_geo2.so; do
  for gb in 768 1024 1536 2048; do
    echo -n "$lib geo_blocks=$gb: "; YSMR_HIP_LIB=$lib YSMR_GEO_BLOCKS=$gb python3 bench.py --config 1 --cpu-sample 0 2>/dev/null | python3 -c "$pick"
  done
done
