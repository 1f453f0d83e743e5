"""Which launches of the threshold kernel take twice as long, and what else is on the device then (kernel trace of a bench run:
rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py ...; argv: DIR)."""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
def short(n):
    for k in ("k_threshold_mfma", "k_batch", "k_clear", "k_windows", "k_residue_frames", "k_rank", "k_geometry", "k_compact", "k_bgrid", "k_tracker_reset"):
        if k in n:
            return k
    return n[:40]
ks = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r["Queue_Id"]) for r in rows)
thr = [(s, e) for s, e, n, q in ks if n == "k_threshold_mfma"]
d = sorted((e - s) / 1e3 for s, e in thr)
print(f"{len(thr)} threshold launches: median {d[len(d) // 2]:.1f} us, max {d[-1]:.1f}; slower than 1.3 x the median: {sum(x > 1.3 * d[len(d) // 2] for x in d)}")
for i, (s, e) in enumerate(thr):
    if (e - s) / 1e3 > 1.3 * d[len(d) // 2]:
        beside = [(n, (a - s) / 1e3, (b - a) / 1e3, q) for a, b, n, q in ks if a < e and b > s and n != "k_threshold_mfma"]
        print(f"launch {i}: {(e - s) / 1e3:.1f} us; beside it (kernel, start relative to the launch's, duration, queue):")
        for n, off, dur, q in beside:
            print(f"     {n:28s} {off:9.1f} {dur:9.1f}  q{q}")
