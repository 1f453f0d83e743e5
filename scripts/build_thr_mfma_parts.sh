#!/bin/bash
# Builds scripts/var_tm_<PART>.so: the library with one part of k_threshold_mfma deleted (see thr_mfma_parts.sh).
cd "$(dirname "$0")/../ysmr_amd/csrc" || exit 1
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I../../include -Wno-unused-function"
for v in NOBLUR NOFILTER NOSTORE NOLOAD NOREFINE; do
  defs=""; for part in ${v//_/ }; do case $part in HALF) defs="$defs -DTM_MAX_PANEL_N=624 -DTM_WAVES_N=8";; THIRD) defs="$defs -DTM_MAX_PANEL_N=416 -DTM_WAVES_N=6";; *) defs="$defs -DTM_DBG_$part";; esac; done
  /opt/rocm/bin/hipcc $FLAGS $defs -c thr_mfma.hip -o /tmp/thr_mfma_$v.o || exit 1
  objs=""; for o in common detect meangray track rows select evaluate ingest; do objs="$objs $o.o"; done
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../../scripts/var_tm_$v.so $objs /tmp/thr_mfma_$v.o || exit 1
done
ls -la ../../scripts/var_tm_*.so
