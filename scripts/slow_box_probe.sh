#!/bin/bash
# Is this box one of those where bench.py runs at ~100 k frames/s?  If so: a kernel trace of the same command and what
# scripts/link_gaps.py reads from it (queues, overlap of the two streams).
cd $GRAFT_REPO_ROOT
v=$(python3 bench.py --steps 10 --warmup 3 --cpu-sample 0 2>/dev/null | python3 -c "import json,sys; r=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print(int(r['value']), r['diagnostics']['host_enqueue_ms_per_step'])")
echo "bench: $v"; env | grep -i "HIP_\|ROC\|HSA_\|GPU_" | head -20
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/kt_probe -- python3 $GRAFT_REPO_ROOT/bench.py --steps 10 --warmup 3 --cpu-sample 0 > $GRAFT_REPO_ROOT/gpurun_out/kt_probe.log 2>&1
cd $GRAFT_REPO_ROOT; grep "^{" gpurun_out/kt_probe.log | cut -c1-160; python3 scripts/link_gaps.py gpurun_out/kt_probe; rm -rf gpurun_out/kt_probe
