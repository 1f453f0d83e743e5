#!/bin/bash
# Matrix-pipe threshold kernel by deletion: builds with one part removed (-DTM_DBG_*; results are wrong, only the
# time is of interest) against the shipped kernel.  Build the variants first: scripts/build_thr_mfma_parts.sh (no GPU needed).
R=$GRAFT_REPO_ROOT
echo "complete: "; python3 $R/scripts/bench_threshold.py | tail -1
for v in NOBLUR NOFILTER NOMFMA NOSTORE NOBLUR_NOFILTER; do
  [ -f $R/scripts/var_tm_$v.so ] && { echo "$v: "; YSMR_HIP_LIB=$R/scripts/var_tm_$v.so python3 $R/scripts/bench_threshold.py | tail -1; }
done
