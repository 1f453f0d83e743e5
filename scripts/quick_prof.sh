#!/bin/bash
# Bench lines (end to end, detection only) and rocprofv3 kernel stats of both (run through gpurun).
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/quick; rm -rf $O; mkdir -p $O
python $R/bench.py --cpu-sample 0 > $O/bench.json 2> $O/bench.err; cut -c1-140 $O/bench.json
python $R/bench.py --cpu-sample 0 --config 1 > $O/bench_det.json 2>> $O/bench.err; cut -c1-140 $O/bench_det.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -- python3 $R/bench.py --cpu-sample 0 > $O/ktrace.log 2>&1
python3 $R/scripts/kstats.py $O/ktrace 24
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace_det -- python3 $R/bench.py --cpu-sample 0 --config 1 > $O/ktrace_det.log 2>&1
python3 $R/scripts/kstats.py $O/ktrace_det 20
