run() { python bench.py --cpu-sample 0 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['value']), round(d['roofline']['avg_launch_ms']*1e3))"; }
run base
YSMR_THR_BLOCKS=512 run thr512
YSMR_THR_BLOCKS=640 run thr640
YSMR_THR_BLOCKS=1024 run thr1024
YSMR_SPARSE_BLOCKS=512 run sparse512
YSMR_SPARSE_BLOCKS=384 run sparse384
YSMR_GEO_BLOCKS=1024 run geo1024
YSMR_GEO_BLOCKS=768 run geo768
YSMR_COLLECT_BLOCKS=512 run collect512
YSMR_THR_BLOCKS=512 YSMR_SPARSE_BLOCKS=512 YSMR_GEO_BLOCKS=1024 YSMR_COLLECT_BLOCKS=512 run all_small
