#!/bin/bash
# SQ counters of k_batch (per launch of 256 frames) inside the running pipeline: separate rocprofv3 --pmc passes over
# bench.py, no tracing domains.  Output: gpurun_out/r04_pmc_batch_link.log
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/pmc_kb; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY --output-format csv -d $O/a -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $O/a.log 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA --output-format csv -d $O/b -- python3 $R/bench.py --steps 3 --warmup 1 --cpu-sample 0 > $O/b.log 2>&1
python3 - <<PY
import csv, glob, collections
acc = collections.defaultdict(list)
for f in glob.glob("$O/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_batch" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("k_batch, per launch of 256 frames (mean over %d launches):" % max(len(v) for v in acc.values()))
for k in sorted(acc): print(f"  {k:24s} {sum(acc[k]) / len(acc[k]):14.0f}   per frame {sum(acc[k]) / len(acc[k]) / 256:10.1f}")
PY
rm -rf $O
