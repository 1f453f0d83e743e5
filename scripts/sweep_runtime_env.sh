#!/bin/bash
# End-to-end frames/s against HIP runtime switches that touch how a launch reaches the GPU (kernel arguments, fences,
# signals).  The link is one short kernel per frame, so the launch path is on the critical path.  Run through gpurun.
R=$GRAFT_REPO_ROOT
run() { echo -n "$*: "; timeout -k 5 90 env "$@" python3 $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read()); g = d['diagnostics']
print(round(d['value']), 'frames/s  link', round(g['link_us_per_frame']['avg'], 2), 'us/frame  host issue', round(g['link_host_issue_us_per_frame']['p50'], 2), 'us/frame (p50)')"; echo; }
run A=0
run HIP_FORCE_DEV_KERNARG=0
run AMD_OPT_FLUSH=0
# (ROC_SYSTEM_SCOPE_SIGNAL=0 hangs the process: the host waits for signals it never sees)
run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=1
run ROC_USE_FGS_KERNARG=0
run ROC_USE_FGS_KERNARG=1
run GPU_FLUSH_ON_EXECUTION=1
run ROC_ACTIVE_WAIT_TIMEOUT=0
run ROC_AQL_QUEUE_SIZE=4096
run GPU_MAX_HW_QUEUES=2
run GPU_MAX_HW_QUEUES=8
run A=0
