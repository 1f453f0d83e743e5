#!/bin/bash
# select_tracks: timing line + rocprofv3 kernel stats (run through gpurun).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/select; rm -rf $O; mkdir -p $O
python $R/tests/tools/bench_select.py > $O/bench.json 2> $O/bench.err; tail -c 700 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -- python3 $R/tests/tools/bench_select.py --reps 5 --cpu-tracks 0 > $O/ktrace.log 2>&1
python3 $R/scripts/kstats.py $O/ktrace 24
