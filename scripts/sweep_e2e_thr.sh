R=$GRAFT_REPO_ROOT
for tb in 0 640 512 384; do
  echo -n "thr_blocks=$tb: e2e "
  YSMR_HIP_LIB=$R/scripts/var_tuning.so YSMR_THR_BLOCKS=$tb python $R/bench.py --cpu-sample 0 2>/dev/null | grep -o '"value": [0-9.]*' | tr '\n' ' '
  echo -n " detect-only "
  YSMR_HIP_LIB=$R/scripts/var_tuning.so YSMR_THR_BLOCKS=$tb python $R/bench.py --cpu-sample 0 --config 1 2>/dev/null | grep -o '"value": [0-9.]*\|"frac": [0-9.]*' | tr '\n' ' '; echo
done
