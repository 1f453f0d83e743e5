#!/bin/bash
# detection alone (bench.py --config 1) and the pipeline once per library given on the command line, two rounds
cd $GRAFT_REPO_ROOT
pick='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); d=r["diagnostics"]; print("%.1f k frames/s, threshold %.1f us, components %.1f us" % (r["value"]/1e3, d["threshold_us_per_batch"]["avg"], d["components_us_per_batch"]["avg"]))'
for rep in 1 2; do
  for lib in "$@"; do
    echo -n "$lib config 1: "; YSMR_HIP_LIB=$lib python3 bench.py --config 1 --cpu-sample 0 2>/dev/null | python3 -c "$pick"
    echo -n "$lib pipeline: "; YSMR_HIP_LIB=$lib python3 bench.py --steps 10 --warmup 3 --cpu-sample 0 2>/dev/null | python3 -c "$pick"
  done
done
