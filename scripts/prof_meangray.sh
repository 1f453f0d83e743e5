#!/bin/bash
# Bench line + rocprofv3 kernel stats of the mean-gray threshold branch (run through gpurun).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/meangray; rm -rf $O; mkdir -p $O
python $R/bench.py --adt -1 --cpu-sample 60 > $O/bench.json 2> $O/bench.err; tail -c 900 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -- python3 $R/bench.py --adt -1 --cpu-sample 0 --steps 3 > $O/ktrace.log 2>&1
python3 $R/scripts/kstats.py $O/ktrace | head -30
