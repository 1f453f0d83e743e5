#!/bin/bash
# Build variants of detect.hip (extra -D flags) into scripts/var_<name>.so and time k_windows / k_clear in the
# detection-only bench (run the build part in the container, the timing part through gpurun).
R=${GRAFT_REPO_ROOT:-/root/repo}; C=$R/ysmr_amd/csrc
if [ "$1" = build ]; then
  shift
  while [ $# -gt 1 ]; do
    name=$1; flags=$2; shift 2
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -fno-fast-math -fno-slp-vectorize -I$R/include $flags -c $C/detect.hip -o /tmp/var_$name.o &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o $R/scripts/var_$name.so /tmp/var_$name.o $C/common.o $C/meangray.o $C/track.o $C/rows.o $C/select.o $C/evaluate.o $C/ingest.o $C/thr_mfma.o && echo built $name
  done
  exit 0
fi
cd /tmp && export TMPDIR=/tmp
for so in $R/scripts/var_*.so; do
  name=$(basename $so .so); O=$R/gpurun_out/winvar/$name; rm -rf $O; mkdir -p $O
  YSMR_HIP_LIB=$so rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --cpu-sample 0 --config 1 > $O.log 2>&1
  echo "== $name: $(grep -o '"value": [0-9.]*' $O.log | head -1)"
  python3 $R/scripts/kstats.py $O 6 | grep -E "k_windows|k_clear|k_geometry" | cut -c1-150
done
