#!/bin/bash
# BASELINE configs[4] (3840x2160, ~5000 blobs): bench line + rocprofv3 kernel stats (run through gpurun).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/k4; rm -rf $O; mkdir -p $O
A="--height 2160 --width 3840 --blobs 5000 --frames 64 --batch ${BATCH:-8} --max-det 8192 --capacity 8192 --cpu-sample 0"
python $R/bench.py --steps 3 $A > $O/bench.json 2> $O/bench.err; cut -c1-160 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -- python3 $R/bench.py --steps 2 $A > $O/ktrace.log 2>&1
python3 $R/scripts/kstats.py $O/ktrace 22
