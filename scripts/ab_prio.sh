#!/bin/bash
# A/B of k_frame's wave priority (s_setprio 0..3) on one box: build scripts/var_prio<N>.so first (track.hip with the level
# replaced, linked like scripts/build_tuning.sh); end-to-end frames/s, the threshold kernel's time beside the link, the link's
# time per frame.  Result: profiles/r03_ab_link_priority.log.
R=$GRAFT_REPO_ROOT
for i in 1 2 3; do
  for v in prio3 prio2 prio1 prio0; do
    echo -n "$v: "; YSMR_HIP_LIB=$R/scripts/var_$v.so timeout -k 5 120 python3 $R/bench.py --cpu-sample 0 --steps 20 --warmup 5 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); g = d['diagnostics']
print(round(d['value']), 'frames/s  threshold', round(g['threshold_us_per_batch']['avg'], 1), 'us  frac', round(d['roofline']['frac'], 4), ' link', round(g['link_us_per_frame']['avg'], 2), 'us/frame')"
  done
done
