#!/bin/bash
# bench.py (driver arguments) once per library given on the command line, twice round: value, threshold kernel, chains
cd $GRAFT_REPO_ROOT
pick='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); d=r["diagnostics"]; print("%.1f k frames/s, threshold %.1f us (frac %.3f), components %.1f us, link %.2f us/frame" % (r["value"]/1e3, d["threshold_us_per_batch"]["avg"], r["roofline"]["frac"], d["components_us_per_batch"]["avg"], d["link_us_per_frame"]["avg"]))'
for rep in 1 2; do
  for lib in "$@"; do
    echo -n "$lib: "; YSMR_HIP_LIB=$lib python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python3 -c "$pick"
  done
done
