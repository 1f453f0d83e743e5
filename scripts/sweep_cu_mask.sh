#!/bin/bash
# EXPERIMENT (dropped; needs the YSMR_SIDE_RESERVE hook that was in ysmr_amd/track_eval.py at the time, see DESIGN.md
# section 8): the detection chain, or the threshold kernel alone, on a stream created with
# hipExtStreamCreateWithCUMask so that some compute units stay free for the link's launches.
R=$GRAFT_REPO_ROOT
for cfg in "0 top" "32 top" "64 top" "64 low" "64 spread" "96 top" "32 spread"; do set -- $cfg
  echo -n "reserve=$1 mode=$2: "
  YSMR_SIDE_RESERVE=$1 YSMR_SIDE_RESERVE_MODE=$2 python $R/bench.py --cpu-sample 0 2>&1 | grep -o '"value": [0-9.]*\|Error.*' | head -2 | tr '\n' ' '; echo
done
