#!/bin/bash
# PMC counters for the threshold kernel (separate passes; never combined with tracing domains)
R=$GRAFT_REPO_ROOT; tag=$1
cd /tmp && export TMPDIR=/tmp
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM" "FETCH_SIZE" "WRITE_SIZE" "GRBM_GUI_ACTIVE"; do
  n=$(echo $set | cut -d' ' -f1)
  rocprofv3 --pmc $set --output-format csv -d $R/gpurun_out/pmc_${tag}_$n -- python3 $R/scripts/bench_threshold.py --reps 1 > $R/gpurun_out/pmc_${tag}_$n.log 2>&1
done
python3 - <<PY
import csv, glob, collections
for d in sorted(glob.glob("$R/gpurun_out/pmc_${tag}_*/")):
    for f in glob.glob(d + "*/*counter_collection.csv"):
        acc = collections.defaultdict(list)
        for r in csv.DictReader(open(f)):
            if "k_threshold" in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k, v in acc.items():
            print(f"{k:28s} n={len(v):3d} mean={sum(v)/len(v):.4g}")
PY
