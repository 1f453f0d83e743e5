#!/bin/bash
# SQ counters of k_threshold_mfma alone at the bench's launch shape (256 frames, the grid beside the batch link), three passes.
# usage (on the GPU box): scripts/pmc_thr_mfma.sh <out-dir>
R=$GRAFT_REPO_ROOT; O=${1:-$R/gpurun_out/pmc_thr}; rm -rf $O; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
B="python3 $R/scripts/bench_threshold.py --reps 1 --real --variant 0 --beside --batch 256"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/a -- $B > $O/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR --output-format csv -d $O/b -- $B > $O/b.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU --output-format csv -d $O/c -- $B > $O/c.log 2>&1
python3 - <<PY
import csv, glob, collections, json
acc = collections.defaultdict(list)
for f in glob.glob("$O/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "k_threshold_mfma" in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {k: sum(v) / len(v) for k, v in acc.items()}
out["launches_seen"] = max(len(v) for v in acc.values()) if acc else 0
json.dump(out, open("$O/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
