#!/bin/bash
# scripts/var_stamps_<NAME>.so: the stamps build of the library with extra defines for thr_mfma.hip (by-deletion builds seen
# through the device stamps).  usage: scripts/build_stamps_variant.sh NAME -DTM_DBG_NOSTORE ...
R=${GRAFT_REPO_ROOT:-/root/repo}; C=$R/ysmr_amd/csrc; T=/tmp/stampbuild; name=$1; shift
[ -f $T/track.o ] || { echo "run scripts/build_stamps.sh first"; exit 1; }
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I$R/include -DYSMR_STAMPS "$@" -c $C/thr_mfma.hip -o /tmp/thr_mfma_st_$name.o || exit 1
objs=""; for f in common detect meangray track rows select evaluate ingest; do objs="$objs $T/$f.o"; done
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/scripts/var_stamps_$name.so $objs /tmp/thr_mfma_st_$name.o && echo built scripts/var_stamps_$name.so
