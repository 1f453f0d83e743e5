#!/bin/bash
# usage (GPU box): scripts/r04_first_batch.sh -- the batch link's first numbers: bench lines, then a kernel trace
R=$GRAFT_REPO_ROOT
cd $R
for args in "--blobs 450 --capacity 512" "--blobs 450 --capacity 512 --max-det 1024" "--blobs 450 --capacity 2048"; do
  echo "== $args"; timeout -k 10 300 python3 bench.py --steps 5 --warmup 2 --cpu-sample 0 $args 2>&1 | tail -1 | cut -c1-2400
done
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_b450 -- python3 $R/bench.py --steps 2 --warmup 1 --cpu-sample 0 --blobs 450 --capacity 512 > $R/gpurun_out/prof_b450.log 2>&1
cd $R && python3 scripts/kstats.py b450 16
