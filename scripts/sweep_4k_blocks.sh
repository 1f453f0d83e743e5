#!/bin/bash
# 4K configuration: end-to-end frames/s against the resident grid of the threshold kernel (tuning build of the
# library: scripts/win_variants.sh build tuning "-DYSMR_TUNING"; run through gpurun).
R=$GRAFT_REPO_ROOT
A="--height 2160 --width 3840 --blobs 5000 --frames 64 --batch 16 --max-det 8192 --capacity 8192 --cpu-sample 0 --steps 3"
for tb in 0 760 744 704 640 512 384; do
  echo -n "thr_blocks=$tb: "
  YSMR_HIP_LIB=$R/scripts/var_tuning.so YSMR_THR_BLOCKS=$tb python $R/bench.py $A 2>/dev/null | grep -o '"value": [0-9.]*'
done
