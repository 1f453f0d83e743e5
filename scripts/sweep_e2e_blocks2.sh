R=$GRAFT_REPO_ROOT
for cfg in "0 0" "1024 0" "1024 768" "1024 512" "1280 768"; do set -- $cfg
  echo -n "window_blocks=$1 geo_blocks=$2: e2e "
  YSMR_HIP_LIB=$R/scripts/var_tuning.so YSMR_COLLECT_BLOCKS=$1 YSMR_GEO_BLOCKS=$2 python $R/bench.py --cpu-sample 0 2>/dev/null | grep -o '"value": [0-9.]*' | tr '\n' ' '
  echo -n " detect-only "
  YSMR_HIP_LIB=$R/scripts/var_tuning.so YSMR_COLLECT_BLOCKS=$1 YSMR_GEO_BLOCKS=$2 python $R/bench.py --cpu-sample 0 --config 1 2>/dev/null | grep -o '"value": [0-9.]*'
done
