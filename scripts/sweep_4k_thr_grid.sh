#!/bin/bash
# configs[4] against the matrix-pipe threshold kernel's grid beside the two-launch link (round 5's kernel); tuning build
cd $GRAFT_REPO_ROOT
pick='import json,sys
r=json.loads([l for l in sys.stdin if l.startswith(chr(123))][-1]); d=r["diagnostics"]
print(int(r["value"]), "frames/s; threshold", round(d["threshold_us_per_batch"]["avg"],1), "us per batch (frac", round(r["roofline"]["frac"],3), "), components", round(d["components_us_per_batch"]["avg"],1), ", link us/frame", round(d["link_us_per_frame"]["avg"],2))'
for rep in 1 2; do for b in 96 128 160 192 224 248; do
  echo -n "threshold kernel on $b workgroups: "
  YSMR_HIP_LIB=scripts/var_tuning.so YSMR_THR_BLOCKS=$b python3 bench.py --config 4 --cpu-sample 0 2>/dev/null | python3 -c "$pick"
done; done
