#!/bin/bash
# per-kernel averages (rocprofv3 --kernel-trace --stats) of the detection-only bench, once per library given: scripts/kstats_libs.sh lib1 lib2 ...
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
for so in "$@"; do
  name=$(basename $so .so); O=$R/gpurun_out/kstats_libs/$name; rm -rf $O; mkdir -p $O
  YSMR_HIP_LIB=$R/$so rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 $R/bench.py --cpu-sample 0 --config 1 --steps 10 --warmup 3 > $O.log 2>&1
  echo "== $name: $(grep -o '"value": [0-9.]*' $O.log | head -1)"
  python3 $R/scripts/kstats.py $O 9 | cut -c1-150
done
