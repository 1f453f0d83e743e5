#!/bin/bash
# usage (on the GPU box, via gpurun): scripts/prof.sh <tag> [bench args...]
# runs the GPU tests, then bench.py under rocprofv3 --kernel-trace --stats; outputs under gpurun_out/
tag=$1; shift
R=$GRAFT_REPO_ROOT
python -m pytest tests -m gpu -q -x > $R/gpurun_out/test_$tag.log 2>&1; tail -3 $R/gpurun_out/test_$tag.log
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$tag -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 "$@" > $R/gpurun_out/prof_$tag.log 2>&1
tail -1 $R/gpurun_out/prof_$tag.log | cut -c1-330
