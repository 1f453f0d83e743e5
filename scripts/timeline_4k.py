"""Who runs when (written for BASELINE configs[4], works for any bench.py run): from a rocprofv3 --kernel-trace csv, the kernels of the last step as one
timeline (start, duration, queue), the busy time of the link kernels, of the detection kernels, and of both at once.

    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 bench.py --config 4 --steps 3 --cpu-sample 0
    python3 scripts/timeline_4k.py DIR [n_lines]
"""
import csv, glob, sys, collections

LINK = ("k_link", "k_track", "k_grid_build", "k_rowmin", "k_frame")


def union(iv):
    iv = sorted(iv)
    out = []
    for s, e in iv:
        if out and s <= out[-1][1]:
            out[-1][1] = max(out[-1][1], e)
        else:
            out.append([s, e])
    return out


def length(iv):
    return sum(e - s for s, e in iv)


def intersect(a, b):
    i = j = 0
    tot = 0
    while i < len(a) and j < len(b):
        s, e = max(a[i][0], b[j][0]), min(a[i][1], b[j][1])
        if s < e:
            tot += e - s
        if a[i][1] < b[j][1]:
            i += 1
        else:
            j += 1
    return tot


def main():
    f = sorted(glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True))[-1]
    lines = int(sys.argv[2]) if len(sys.argv) > 2 else 120
    rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0],
             r.get("Queue_Id", "?")) for r in csv.DictReader(open(f))]
    rows.sort()
    # the last 40 % of the run (steady state, no warm-up)
    lk_all = [r for r in rows if r[2].startswith(("k_link", "k_frame"))]
    t0, t1 = lk_all[0][0], lk_all[-1][1]
    lo, hi = t0 + (t1 - t0) * 5 // 10, t0 + (t1 - t0) * 9 // 10
    win = [r for r in rows if lo <= r[0] <= hi]
    link = union([(s, e) for s, e, n, q in win if n.startswith(LINK)])
    det = union([(s, e) for s, e, n, q in win if not n.startswith(LINK)])
    span = win[-1][1] - win[0][0]
    print(f"window {span / 1e3:.0f} us: link kernels busy {length(link) / 1e3:.0f} us, detection kernels busy {length(det) / 1e3:.0f} us, "
          f"both at once {intersect(link, det) / 1e3:.0f} us, neither {(span - length(union(link + det))) / 1e3:.0f} us")
    per = collections.defaultdict(list)
    for s, e, n, q in win:
        per[n].append(e - s)
    for n, d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
        print(f"  {n:28s} calls {len(d):5d}  total {sum(d) / 1e3:8.0f} us  mean {sum(d) / len(d) / 1e3:7.2f} us")
    # gaps inside the link chain (end of one link kernel to the start of the next)
    lk = [(s, e, n) for s, e, n, q in win if n.startswith(LINK)]
    gaps = collections.defaultdict(list)
    for a, b in zip(lk, lk[1:]):
        gaps[a[2] + " -> " + b[2]].append(b[0] - a[1])
    for k, g in gaps.items():
        g.sort()
        print(f"  gap {k:32s} n {len(g):4d}  p50 {g[len(g) // 2] / 1e3:6.2f}  mean {sum(g) / len(g) / 1e3:6.2f}  max {g[-1] / 1e3:7.2f} us")
    # link kernels by the detection kernel that was running when they started
    dk = sorted((s, e, n) for s, e, n, q in win if not n.startswith(LINK) and n.startswith("k_"))
    by = collections.defaultdict(list)
    for s, e, n, q in win:
        if n in ("k_link", "k_track", "k_frame", "k_rowmin"):
            co = [d[2] for d in dk if d[0] <= s < d[1]]
            by[(n, co[0] if co else "-")].append(e - s)
    print("  link kernel beside ...: calls, median, mean, max us, share of the link chain's time")
    tot = sum(sum(v) for v in by.values())
    for (n, co), v in sorted(by.items()):
        v.sort()
        print(f"    {n:8s} {co:20s} {len(v):4d} {v[len(v) // 2] / 1e3:7.2f} {sum(v) / len(v) / 1e3:7.2f} {v[-1] / 1e3:8.2f}  {100 * sum(v) / tot:5.1f} %")
    base = win[0][0]
    for s, e, n, q in win[:lines]:
        print(f"{(s - base) / 1e3:9.2f} +{(e - s) / 1e3:7.2f}  q{q}  {n}")


main()
