R=$GRAFT_REPO_ROOT
for b in 8 16 32 64; do
  echo -n "batch=$b: "
  python $R/bench.py --height 2160 --width 3840 --blobs 5000 --frames 64 --batch $b --max-det 8192 --capacity 8192 --cpu-sample 0 --steps 3 2>/dev/null | grep -o '"value": [0-9.]*'
done
for md in 6144; do
  echo -n "batch=16 max_det=capacity=$md: "
  python $R/bench.py --height 2160 --width 3840 --blobs 5000 --frames 64 --batch 16 --max-det $md --capacity $md --cpu-sample 0 --steps 3 2>/dev/null | grep -o '"value": [0-9.]*'
done
