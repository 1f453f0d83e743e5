#!/bin/bash
# Does a smaller link kernel leave room for the matrix-pipe threshold kernel?  k_frame's grid and LDS follow the table sizes
# (capacity / max_det 2048: 512 workgroups of 59 KB, two per compute unit; 1024: 256 workgroups of ~30 KB, one per unit).
R=$GRAFT_REPO_ROOT
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; g=d['diagnostics']
print('%-44s %9.0f frames/s   %-18s %6.1f us/launch  frac %.3f   link %5.2f us/frame' % (sys.argv[1], d['value'], r['kernel'], r['avg_launch_ms']*1e3, r['frac'], (g['link_us_per_frame'] or {'avg':0})['avg']))" "$1"; }
for cap in 2048 1024; do
  for m in beside-strip beside-mfma; do
    YSMR_THRESHOLD_MODE=$m python3 $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 --capacity $cap --max-det $cap 2>/dev/null | show "capacity $cap $m"
  done
  [ -f $R/scripts/var_tm_HALF.so ] && YSMR_HIP_LIB=$R/scripts/var_tm_HALF.so YSMR_THRESHOLD_MODE=beside-mfma python3 $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 --capacity $cap --max-det $cap 2>/dev/null | show "capacity $cap beside-mfma, half-row 8-wave build"
done
