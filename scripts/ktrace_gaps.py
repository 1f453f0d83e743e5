"""Per-kernel duration and launch-to-launch gaps from a rocprofv3 --kernel-trace csv."""
import csv, glob, sys, collections
f = sorted(glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
prev_end = None; prev = None
for r in rows:
    n = r['Kernel_Name'].split('(')[0][:40]
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    dur[n].append(e - s)
    if prev_end is not None and prev == n: gap[n].append(s - prev_end)
    prev_end, prev = e, n
import statistics as st
for n in dur:
    g = gap.get(n, [0])
    print(f"{n:40s} calls={len(dur[n]):6d} dur med={st.median(dur[n])/1e3:7.2f} mean={st.mean(dur[n])/1e3:7.2f} us   gap med={st.median(g)/1e3:6.2f} mean={st.mean(g)/1e3:6.2f} us")
