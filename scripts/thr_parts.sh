#!/bin/bash
# Threshold kernel by deletion: builds with one part of a step removed (-DSTRIP_DBG_*; results are wrong, only
# the time is of interest) against the shipped kernel.
R=$GRAFT_REPO_ROOT
echo -n "complete: "; python $R/scripts/bench_threshold.py | tail -1
for v in NOBLUR NOLOAD NOEXCH NOCOL NOROW NOSTORE NOFILT NOALL; do echo -n "$v: "; YSMR_HIP_LIB=$R/scripts/var_$v.so python $R/scripts/bench_threshold.py | tail -1; done
