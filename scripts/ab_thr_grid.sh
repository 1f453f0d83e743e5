#!/bin/bash
# bench.py (driver arguments) with the matrix-pipe threshold kernel on grids of 240..256 workgroups beside the batch link
# (scripts/var_tuning.so: YSMR_THR_BLOCKS), and the in-tree library as it ships.
cd $GRAFT_REPO_ROOT
pick='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); d=r["diagnostics"]; print("%.1f k frames/s, threshold %.1f us (frac %.3f), components %.1f us, link %.2f us/frame" % (r["value"]/1e3, d["threshold_us_per_batch"]["avg"], r["roofline"]["frac"], d["components_us_per_batch"]["avg"], d["link_us_per_frame"]["avg"]))'
echo -n "shipped: "; python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python3 -c "$pick"
for g in 248 192 160 128 96; do
  echo -n "grid $g: "; YSMR_HIP_LIB=scripts/var_tuning.so YSMR_THR_BLOCKS=$g python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python3 -c "$pick"
done
echo -n "old library: "; YSMR_HIP_LIB=scripts/var_thr_old.so python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python3 -c "$pick"
echo -n "shipped: "; python3 bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python3 -c "$pick"
