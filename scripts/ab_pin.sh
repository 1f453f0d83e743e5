#!/bin/bash
# bench.py with and without the process pinned to the CPUs next to its GPU (YSMR_BENCH_PIN=1), alternating on one box:
# value, the host's enqueue time per step, the link's time per frame
cd $GRAFT_REPO_ROOT
pick='import json,sys
for l in sys.stdin:
    if l.startswith("{"):
        r=json.loads(l); d=r["diagnostics"]; print("%.1f k frames/s, host enqueue %.2f ms of %.2f ms per step, link %.2f us/frame, pinned %s, load %.0f" % (r["value"]/1e3, d["host_enqueue_ms_per_step"], r["ms_per_step"], d["link_us_per_frame"]["avg"], d["pinned_to_gpu_numa_node"], d["host_load_1m"]))'
python3 - <<'PY'
import glob
for d in sorted(glob.glob("/sys/bus/pci/devices/*/local_cpulist")):
    dev = d.rsplit("/", 2)[1]
    try:
        cls = open(d.replace("local_cpulist", "class")).read().strip()
        if cls.startswith("0x0302") or cls.startswith("0x0380") or cls.startswith("0x1200"):
            print(dev, cls, "local cpus", open(d).read().strip(), "numa", open(d.replace("local_cpulist", "numa_node")).read().strip())
    except OSError:
        pass
PY
nproc; cat /sys/fs/cgroup/cpu.max 2>/dev/null
for rep in 1 2 3; do
  echo -n "unpinned: "; python3 bench.py --steps 10 --warmup 3 --cpu-sample 0 2>/dev/null | python3 -c "$pick"
  echo -n "pinned:   "; YSMR_BENCH_PIN=1 python3 bench.py --steps 10 --warmup 3 --cpu-sample 0 2>/dev/null | python3 -c "$pick"
done
