"""Batch-boundary gaps and k_frame duration histogram from a rocprofv3 kernel trace of bench.py."""
import csv, glob, re, sys, statistics as st, collections
f = sorted(glob.glob(sys.argv[1] + '/**/*_kernel_trace.csv', recursive=True))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
kf = [(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in rows if 'k_frame' in r['Kernel_Name']]
gaps = [kf[i+1][0]-kf[i][1] for i in range(len(kf)-1)]
durs = [e-s for s,e in kf]
k0 = len(kf)//3
print("k_frame dur med %.2f mean %.2f us" % (st.median(durs)/1e3, st.mean(durs)/1e3))
big = [g for g in gaps[k0:] if g >= 20000]
print("timed span %.2f ms: k_frame %.2f ms, small gaps %.2f ms, %d big gaps %.2f ms (med %.0f us)" % ((kf[-1][1]-kf[k0][0])/1e6, sum(durs[k0:])/1e6,
      sum(g for g in gaps[k0:] if g < 20000)/1e6, len(big), sum(big)/1e6, st.median(big)/1e3 if big else 0))
h = collections.Counter(min(int(d/4000)*4, 60) for d in durs[k0:])
print("dur histogram (us bucket: count):", sorted(h.items()))
# what runs on the other queue during slow k_frames
slow = [(s,e) for s,e in kf[k0:] if e-s > 20000]
names = collections.Counter()
for s,e in slow:
    for r in rows:
        if 'k_frame' in r['Kernel_Name']: continue
        rs, re_ = int(r['Start_Timestamp']), int(r['End_Timestamp'])
        if rs < e and re_ > s: m_ = re.search(r'(k_\w+|fillBuffer\w*|copyBuffer)', r['Kernel_Name']); names[m_.group(1) if m_ else r['Kernel_Name'][:30]] += 1
print("kernels overlapping slow k_frames (> 20 us):", len(slow), names.most_common(12))
