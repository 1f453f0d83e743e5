#!/bin/bash
# Stamps build of the library (device-side s_memtime / s_memrealtime stamps read by scripts/*stamps*.py,
# link_phases.py, link_timeline.py, ring_gaps.py): scripts/var_stamps.so, selected with YSMR_HIP_LIB.  Run in the
# build container (hipcc cross-compiles).
R=${GRAFT_REPO_ROOT:-/root/repo}; C=$R/ysmr_amd/csrc; T=/tmp/stampbuild; mkdir -p $T
for f in common detect thr_mfma meangray track rows select evaluate ingest; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I$R/include -DYSMR_STAMPS $EXTRA -c $C/$f.hip -o $T/$f.o || exit 1
done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/scripts/var_stamps.so $T/*.o && echo built scripts/var_stamps.so
