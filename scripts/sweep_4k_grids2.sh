#!/bin/bash
# configs[4] against the resident grids of k_windows / k_geometry, with the matrix-pipe threshold kernel on 128 workgroups
# (the shipped choice beside the two-launch link); tuning build
cd $GRAFT_REPO_ROOT
pick='import json,sys
r=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); d=r["diagnostics"]
print(int(r["value"]), "threshold", round(d["threshold_us_per_batch"]["avg"],1), "components", round(d["components_us_per_batch"]["avg"],1), "link us/frame", round(d["link_us_per_frame"]["avg"],2))'
for w in 0 512 1024; do for g in 0 512 768; do
  echo -n "k_windows blocks ${w} (0: default 2048), k_geometry blocks ${g} (0: default 1536): "
  YSMR_HIP_LIB=scripts/var_tuning.so YSMR_COLLECT_BLOCKS=$w YSMR_GEO_BLOCKS=$g python3 bench.py --config 4 --cpu-sample 0 2>/dev/null | python3 -c "$pick"
done; done
for b in 8 32; do echo -n "detect batch $b: "; python3 bench.py --config 4 --batch $b --cpu-sample 0 2>/dev/null | python3 -c "$pick"; done
