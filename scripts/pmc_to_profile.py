"""gpurun_out/round/pmc_summary.json (scripts/profile_round.sh) -> profiles/threshold_pmc.json: HBM bytes per launch of
the threshold kernel, FETCH_SIZE / WRITE_SIZE (KB) corrected by the calibration copy (580 MiB read + 580 MiB written,
scripts/ubench/copy_calib.hip), as MI355X_MICROARCH.md's HBM section prescribes.  bench.py copies hbm_bytes_per_launch
into roofline.traffic.   python scripts/pmc_to_profile.py [round-dir] [note]"""
import json, os, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[1] if len(sys.argv) > 1 else os.path.join(root, "gpurun_out", "round")
s = json.load(open(os.path.join(src, "pmc_summary.json")))
true = 580 * 2 ** 20
rf = s["calib_FETCH_SIZE"] * 1024 / true          # counter bytes per true byte
wf = s["calib_WRITE_SIZE"] * 1024 / true
rd, wr = s["threshold_FETCH_SIZE"] * 1024 / rf, s["threshold_WRITE_SIZE"] * 1024 / wf
B, H, W = int(os.environ.get("PMC_BATCH", "256")), 922, 1228      # (the batch scripts/profile_round.sh passes to bench_threshold.py)
out = {"kernel": "k_threshold_strip", "batch": B, "height": H, "width": W,
       "FETCH_SIZE_KB_raw": s["threshold_FETCH_SIZE"], "WRITE_SIZE_KB_raw": s["threshold_WRITE_SIZE"],
       "calibration": {"kernel": "copy_dword (scripts/ubench/copy_calib.hip): 580 MiB read + 580 MiB written, one dword per lane",
                       "FETCH_SIZE_KB": s["calib_FETCH_SIZE"], "WRITE_SIZE_KB": s["calib_WRITE_SIZE"],
                       "fetch_counter_over_true_bytes": rf, "write_counter_over_true_bytes": wf},
       "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
       "algorithmic_bytes_per_launch": 2 * B * H * W,
       "sq": {k[len("threshold_"):]: v for k, v in s.items() if k.startswith("threshold_SQ_")},
       "note": sys.argv[2] if len(sys.argv) > 2 else ""}
json.dump(out, open(os.path.join(root, "profiles", "threshold_pmc.json"), "w"), indent=1)
print(f"k_threshold_strip: read {rd / 1e6:.1f} MB + written {wr / 1e6:.1f} MB = {(rd + wr) / (2 * B * H * W):.2f} x algorithmic")
if s.get("mfma_FETCH_SIZE") is not None:
    B = int(os.environ.get("PMC_BATCH_MFMA", "248"))     # (the matrix-pipe kernel's launches: bench.py's batch beside the batch link)
    rd, wr = s["mfma_FETCH_SIZE"] * 1024 / rf, s["mfma_WRITE_SIZE"] * 1024 / wf
    out = {"kernel": "k_threshold_mfma", "batch": B, "height": H, "width": W, "FETCH_SIZE_KB_raw": s["mfma_FETCH_SIZE"],
           "WRITE_SIZE_KB_raw": s["mfma_WRITE_SIZE"], "calibration": out["calibration"],
           "hbm_read_bytes_per_launch": rd, "hbm_write_bytes_per_launch": wr, "hbm_bytes_per_launch": rd + wr,
           "algorithmic_bytes_per_launch": 2 * B * H * W,
           "sq": {k[len("mfma_"):]: v for k, v in s.items() if k.startswith("mfma_SQ_")}, "note": sys.argv[2] if len(sys.argv) > 2 else ""}
    json.dump(out, open(os.path.join(root, "profiles", "threshold_mfma_pmc.json"), "w"), indent=1)
    print(f"k_threshold_mfma:  read {rd / 1e6:.1f} MB + written {wr / 1e6:.1f} MB = {(rd + wr) / (2 * B * H * W):.2f} x algorithmic")
