#!/bin/bash
# Counters of the link kernels (k_link, k_track at BASELINE configs[4]; BENCH_CFG=" " for k_frame at the headline size): is k_track's 12 us a latency chain or
# VALU issue?  One rocprofv3 --pmc pass (no tracing domains) over a short bench run; per-kernel means.  Run through gpurun.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc4k; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/a -- python3 $R/bench.py ${BENCH_CFG:---config 4} --steps 1 --warmup 0 --cpu-sample 0 > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --output-format csv -d $O/b -- python3 $R/bench.py ${BENCH_CFG:---config 4} --steps 1 --warmup 0 --cpu-sample 0 > $O/b.log 2>&1
python3 - <<PY
import csv, glob, collections
for d in ("a", "b"):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob("$O/" + d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            n = r["Kernel_Name"]
            for k in ("k_link", "k_track", "k_frame", "k_threshold_strip", "k_windows"):
                if k in n: acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, c in acc.items():
        print(k, {n: round(sum(v) / len(v)) for n, v in c.items()}, "launches", len(next(iter(c.values()))))
PY
