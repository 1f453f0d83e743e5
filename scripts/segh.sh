#!/bin/bash
for sh in 45 56 67 100 111 122 155; do echo -n "seg_h=$sh: "; YSMR_SEG_H=$sh python $GRAFT_REPO_ROOT/scripts/bench_threshold.py | tail -1; done
