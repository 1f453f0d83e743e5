#!/bin/bash
# bench.py --config C with each library given (`default`: the tree's), same box, alternating: frames/s, the link's and the
# labelling chain's time.  usage: scripts/ab_bench_cfg.sh C lib1 lib2 ...
cd $GRAFT_REPO_ROOT
C=$1; shift
for rep in 1 2; do
for lib in "$@"; do
  unset YSMR_HIP_LIB; [ "$lib" = default ] || export YSMR_HIP_LIB=$lib
  python3 bench.py --config $C --cpu-sample 0 2>/dev/null | python3 -c "
import json, sys
d = json.loads(sys.stdin.readline()); g = d['diagnostics']
lk = (g.get('link_us_per_frame') or {}).get('avg'); cp = (g.get('components_us_per_batch') or {}).get('avg')
print('$lib config $C:', round(d['value']), 'frames/s  link us/frame', lk and round(lk, 1), ' components us/batch', cp and round(cp, 1), ' threshold us', round(d['roofline']['avg_launch_ms'] * 1e3, 1))"
done
done
