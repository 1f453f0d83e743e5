#!/bin/bash
# bench.py --config 4 (4K, ~5000 blobs) with each library given (YSMR_HIP_LIB; `default`: the tree's; `waves:<lib>`: a -DYSMR_TUNING build with YSMR_LINK_MODE=waves, k_track instead of k_track_lanes), same box: the link's us per frame and frames/s
cd $GRAFT_REPO_ROOT
for lib in "$@"; do
  unset YSMR_HIP_LIB YSMR_LINK_MODE
  case "$lib" in default) ;; waves:*) export YSMR_LINK_MODE=waves YSMR_HIP_LIB=${lib#waves:} ;; *) export YSMR_HIP_LIB=$lib ;; esac
  python3 bench.py --config 4 --steps 12 --warmup 3 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline()); g = d['diagnostics']
print('$lib', round(d['value']), 'frames/s  link', round(g['link_us_per_frame']['avg'], 1), 'us/frame  tracks', d['config']['tracks_alive'])"
done
