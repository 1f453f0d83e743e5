#!/bin/bash
# Round profile bundle (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats of
# the same command, PMC passes (each in its own run, no tracing domains) for the threshold kernel
# plus the FETCH/WRITE calibration copy.  Everything lands in gpurun_out/round/.
# Round 5: every call writes into gpurun_out/round_<tag> (tag = $1, default "now"), which the builder deletes locally before the
# call (gpurun MERGES what a call wrote into the local gpurun_out/: round 4's directory had ten runs' files side by side, and
# the copy into profiles/ took the oldest), and the per-kernel summaries are copied to fixed names by kstats.py's rule (the
# newest *_kernel_stats.csv of the directory): profiles/rNN_kernel_stats.csv is $O/kernel_stats.csv, nothing else.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/round_${1:-now}; rm -rf $O; mkdir -p $O
git -C $R rev-parse HEAD > $O/HEAD 2>/dev/null || true
[ -x $R/scripts/ubench/copy_calib ] || make -C $R/scripts/ubench copy_calib >/dev/null
python3 $R/bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; tail -c 600 $O/bench.json
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace -- python3 $R/bench.py --steps 20 --warmup 5 --cpu-sample 0 > $O/ktrace.log 2>&1
if [ -z "$SKIP_PMC" ]; then   # (SKIP_PMC=1: the threshold kernels have not changed since the committed counters)
# (variant 0 = k_threshold_mfma, since round 4 the kernel of the pipeline and of the bench line, on the grid it takes there
#  (--beside: 248 workgroups, and since the end of round 5 248 frames per launch: a frame per workgroup); variant 1 = k_threshold_strip, 256 frames)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_thr_$c -- python3 $R/scripts/bench_threshold.py --reps 1 --real --variant 1 --batch 256 > $O/pmc_thr_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_mfma_$c -- python3 $R/scripts/bench_threshold.py --reps 1 --real --variant 0 --beside --batch 248 --frames 496 > $O/pmc_mfma_$c.log 2>&1
  rocprofv3 --pmc $c --output-format csv -d $O/pmc_cal_$c -- $R/scripts/ubench/copy_calib > $O/pmc_cal_$c.log 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_thr_SQ -- python3 $R/scripts/bench_threshold.py --reps 1 --real --variant 1 --batch 256 > $O/pmc_thr_SQ.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_ACTIVE_INST_VALU --output-format csv -d $O/pmc_mfma_SQ -- python3 $R/scripts/bench_threshold.py --reps 1 --real --variant 0 --beside --batch 248 --frames 496 > $O/pmc_mfma_SQ.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR --output-format csv -d $O/pmc_mfma_SQ2 -- python3 $R/scripts/bench_threshold.py --reps 1 --real --variant 0 --beside --batch 248 --frames 496 > $O/pmc_mfma_SQ2.log 2>&1
python3 - <<PY
import csv, glob, collections, json
O = "$O"
def mean(path, kernel):
    acc = collections.defaultdict(list)
    for f in glob.glob(path + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if kernel in r["Kernel_Name"]:
                acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}
out = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    out["threshold_" + c] = mean(f"{O}/pmc_thr_{c}", "k_threshold").get(c)
    out["mfma_" + c] = mean(f"{O}/pmc_mfma_{c}", "k_threshold").get(c)
    out["calib_" + c] = mean(f"{O}/pmc_cal_{c}", "copy_dword").get(c)
out.update({"threshold_" + k: v for k, v in mean(f"{O}/pmc_thr_SQ", "k_threshold").items()})
out.update({"mfma_" + k: v for k, v in mean(f"{O}/pmc_mfma_SQ", "k_threshold").items()})
out.update({"mfma_" + k: v for k, v in mean(f"{O}/pmc_mfma_SQ2", "k_threshold").items()})
json.dump(out, open(f"{O}/pmc_summary.json", "w"), indent=1)
print(json.dumps(out))
PY
fi
# the other single-GPU configurations of BASELINE.json as bench lines, and the k_windows counters
for c in 0 1 4; do python3 $R/bench.py --config $c --cpu-sample 20 2>> $O/bench.err >> $O/bench_configs.jsonl; done
cut -c1-150 $O/bench_configs.jsonl
rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktrace_det -- python3 $R/bench.py --cpu-sample 0 --config 1 > $O/ktrace_det.log 2>&1
python3 $R/scripts/kstats.py $O/ktrace 30 > $O/kernel_stats.txt; python3 $R/scripts/kstats.py $O/ktrace_det 20 > $O/kernel_stats_detect_only.txt
cp "$(ls -t $O/ktrace/*/*_kernel_stats.csv | head -1)" $O/kernel_stats.csv; cp "$(ls -t $O/ktrace_det/*/*_kernel_stats.csv | head -1)" $O/kernel_stats_detect_only.csv
cat $O/kernel_stats.txt | cut -c1-150
# configs[4]: who runs beside whom (kernel trace), and the link's phases alone / beside detection (device stamps; needs
# scripts/var_stamps.so from scripts/build_stamps.sh)
rocprofv3 --kernel-trace --output-format csv -d $O/kt4k -- python3 $R/bench.py --config 4 --steps 6 --cpu-sample 0 > $O/kt4k.log 2>&1
python3 $R/scripts/timeline_4k.py $O/kt4k 60 > $O/timeline_4k.txt; rm -rf $O/kt4k; head -45 $O/timeline_4k.txt | cut -c1-150
[ -f $R/scripts/var_stamps.so ] && YSMR_HIP_LIB=$R/scripts/var_stamps.so python3 $R/scripts/link_timeline.py > $O/link_timeline_4k.log 2>&1; cat $O/link_timeline_4k.log | grep -v amdgpu
# round 4: what lies between two k_batch launches (the same kernel trace), the phases of the batch link and of the threshold
# kernel's walk (stamps build), file to rows, the two-rank rehearsal on one device
python3 $R/scripts/link_gaps.py $O/ktrace > $O/link_gaps.log 2>&1; cat $O/link_gaps.log
if [ -f $R/scripts/var_stamps.so ]; then
  YSMR_HIP_LIB=$R/scripts/var_stamps.so BL_WAVES=12 python3 $R/scripts/batch_stamps.py 500 768 2048 2>&1 | grep -v amdgpu > $O/batch_link_stamps.log; tail -14 $O/batch_link_stamps.log | cut -c1-170
  YSMR_HIP_LIB=$R/scripts/var_stamps.so python3 $R/scripts/thr_stamps.py 2>&1 | grep -v amdgpu > $O/thr_stamps.log; cat $O/thr_stamps.log | cut -c1-170
fi
python3 $R/scripts/e2e_profile.py > $O/e2e_file_profile.log 2>&1; tail -12 $O/e2e_file_profile.log
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 $R/bench.py --gpus 2 --steps 10 --warmup 3 --dist-backend gloo --device-index 0 --cpu-sample 0 > $O/bench_gloo2.json 2> $O/bench_gloo2.err; cut -c1-260 $O/bench_gloo2.json
rm -rf $O/ktrace/*/*kernel_trace.csv $O/ktrace_det/*/*kernel_trace.csv
