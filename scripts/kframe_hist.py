import csv, glob, sys, numpy as np
f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
d = np.array([int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "k_frame" in r["Kernel_Name"]]) / 1e3
d = d[len(d) // 3:]
print("k_frame calls", len(d), "mean %.2f" % d.mean(), "percentiles 10/50/75/90/95/99:", np.percentile(d, [10, 50, 75, 90, 95, 99]).round(2))
for lo, hi in [(0, 8), (8, 9), (9, 10), (10, 12), (12, 15), (15, 20), (20, 30), (30, 1000)]:
    m = (d >= lo) & (d < hi)
    print(f"  {lo:3d}-{hi:4d} us: {m.sum():5d} calls, {100 * d[m].sum() / d.sum():5.1f} % of the time")
