run() { python bench.py --cpu-sample 0 $2 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), d['config']['tracks_alive'], d['config']['ids_issued'])"; }
run cap2048 ""
run cap1024 "--capacity 1024 --max-det 1024"
run cap768 "--capacity 768 --max-det 768"
run cap1024_md2048 "--capacity 1024 --max-det 2048"
run cap4096 "--capacity 4096 --max-det 2048"
