#!/bin/bash
# The library with its tuning hooks compiled in (-DYSMR_TUNING: resident grids, link mode and table placement from the
# environment) as scripts/var_tuning.so; run in the container, the sweeps that use it through gpurun.
R=${GRAFT_REPO_ROOT:-/root/repo}; C=$R/ysmr_amd/csrc
F="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I$R/include -DYSMR_TUNING $EXTRA"
for f in detect track; do /opt/rocm/bin/hipcc $F -c $C/$f.hip -o /tmp/tuning_$f.o || exit 1; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/scripts/var_tuning.so /tmp/tuning_detect.o /tmp/tuning_track.o $C/common.o $C/thr_mfma.o \
  $C/meangray.o $C/rows.o $C/select.o $C/evaluate.o $C/ingest.o && echo built scripts/var_tuning.so
