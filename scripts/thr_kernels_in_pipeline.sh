#!/bin/bash
# The two threshold kernels inside the pipeline and alone: end-to-end frames/s and the kernel's own time
# between HIP events, for the configuration the metric is quoted on, detection only (configs[1]) and 4K (configs[4]).
# YSMR_THRESHOLD_MODE overrides TrackingPipeline's choice: which kernel (strip = float32 chain, mfma = matrix pipe) and where
# (beside = on the side stream, next to the link; exclusive = on the link stream, between two batches' link chains).
R=$GRAFT_REPO_ROOT
show() { python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); r=d['roofline']; g=d['diagnostics']
print('%-34s %9.0f frames/s   %-18s %6.1f us/launch  frac %.3f   link %5.2f us/frame' % (sys.argv[1], d['value'], r['kernel'], r['avg_launch_ms']*1e3, r['frac'], (g['link_us_per_frame'] or {'avg':0})['avg']))" "$1"; }
for m in beside-strip beside-mfma exclusive-mfma exclusive-strip; do
  YSMR_THRESHOLD_MODE=$m python3 $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | show "configs[2] $m"
done
for m in beside-strip beside-mfma; do
  YSMR_THRESHOLD_MODE=$m python3 $R/bench.py --config 1 --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | show "configs[1] detection only, $m"
done
for m in beside-strip beside-mfma exclusive-mfma; do
  YSMR_THRESHOLD_MODE=$m python3 $R/bench.py --config 4 --steps 10 --warmup 2 --cpu-sample 0 2>/dev/null | show "configs[4] 4K, $m"
done
