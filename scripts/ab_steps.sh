#!/bin/bash
# VERDICT r02 task 1: the driver's command line (--steps 20 --warmup 5) next to the builder's default
# (--steps 5 --warmup 1) on ONE box, alternating, so a difference is the run length's and not the box's.
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/ab_steps.log
: > $O
for i in 1 2 3; do
  echo "== driver  (20/5) run $i" >> $O; python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 --cpu-sample 0 2>/dev/null >> $O
  echo "== default (5/1)  run $i" >> $O; python3 $R/bench.py --gpus 1 --steps 5 --warmup 1 --cpu-sample 0 2>/dev/null >> $O
done
echo "== driver, full line incl. cpu sample" >> $O; python3 $R/bench.py --gpus 1 --steps 20 --warmup 5 >> $O 2>&1
grep -o '"value": [0-9.]*\|^==.*' $O
