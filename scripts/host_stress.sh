#!/bin/bash
# How much host does the pipeline need?  bench.py alone, and beside N busy host processes pinned nowhere in particular
# (what a shared or slower host does to the launch rate), for the library in the tree and scripts/var_old.so
# (round 3 tried replaying a batch.s link chain as a hipGraph: profiles/r03_host_stress_graph_vs_launches.log; hipGraphLaunch of 65 kernel nodes costs the host as much as 65 launches on ROCm 7.2, so the graph build was dropped; var_old.so is whatever older build is being compared).
R=$GRAFT_REPO_ROOT
run() { python3 $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python3 -c "
import json,sys; d=json.loads(sys.stdin.readlines()[-1]); g=d['diagnostics']
print('%8.0f frames/s  link %.2f us/frame  host issue %.2f us/frame  enqueue %.2f ms/step' % (d['value'], g['link_us_per_frame']['avg'], g['link_host_issue_us_per_frame']['avg'], g['host_enqueue_ms_per_step']))"; }
for busy in 0 64 256; do
  pids=""
  for i in $(seq 1 $busy); do ( while :; do :; done ) & pids="$pids $!"; done
  echo "== $busy busy host processes (host has $(nproc) cpus)"
  echo -n "graph replay : "; run
  echo -n "plain launches: "; YSMR_HIP_LIB=$R/scripts/var_old.so run
  [ -n "$pids" ] && kill $pids 2>/dev/null; wait 2>/dev/null
done
