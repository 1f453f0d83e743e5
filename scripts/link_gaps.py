"""Where a batch's time goes on the link stream: rocprofv3 --kernel-trace csv of bench.py -> durations of k_batch, the
gaps between consecutive k_batch launches, and the kernels that start inside those gaps.   usage: link_gaps.py <dir>"""
import csv, glob, sys
import numpy as np
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], r.get("Queue_Id", "")))
rows.sort()
kb = [r for r in rows if "k_batch" in r[2]]
dur = np.array([e - s for s, e, _, _ in kb]) / 1e3
gap = np.array([kb[i + 1][0] - kb[i][1] for i in range(len(kb) - 1)]) / 1e3
print(f"k_batch: {len(kb)} launches, duration mean {dur.mean():.1f} us, p50 {np.median(dur):.1f}, min {dur.min():.1f}, max {dur.max():.1f}")
print(f"gap to the next k_batch: mean {gap.mean():.1f} us, p50 {np.median(gap):.1f}, min {gap.min():.1f}, max {gap.max():.1f}; batches per second of trace {len(kb) / ((kb[-1][1] - kb[0][0]) / 1e9):.0f}")
# what runs in a typical gap
import collections
inside = collections.Counter()
for i in range(len(kb) - 1):
    for s, e, n, q in rows:
        if kb[i][1] <= s < kb[i + 1][0]: inside[n[:40]] += 1
print("kernels that START inside a gap (per gap):", {k: round(v / (len(kb) - 1), 2) for k, v in inside.most_common(8)})
# which hardware queue each kernel ran on, and how much of the link kernels' time a detection kernel overlapped
queues = collections.defaultdict(collections.Counter)
for s_, e_, n_, q_ in rows: queues[q_][n_[:24]] += 1
for q_, c_ in queues.items(): print("queue", q_, dict(c_.most_common(6)))
det = sorted((s_, e_) for s_, e_, n_, q_ in rows if any(k in n_ for k in ("k_threshold", "k_windows", "k_geometry", "k_clear", "k_residue", "k_rank", "k_nested", "k_compact", "k_bgrid")))
busy = 0
j = 0
for s_, e_, _, _ in kb:
    for ds, de in det:
        lo, hi = max(s_, ds), min(e_, de)
        if hi > lo: busy += hi - lo
print(f"detection kernels overlap {busy / 1e3 / max(1, len(kb)):.1f} us of a k_batch launch's {dur.mean():.1f} us on average")
big = np.argsort(gap)[-5:]
for i in big:
    print(f"  gap {gap[i]:.1f} us after launch {i} (duration {dur[i]:.1f})")
