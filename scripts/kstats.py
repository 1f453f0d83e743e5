"""Print the top rows of a rocprofv3 --kernel-trace --stats run.
usage: kstats.py <tag | directory> [rows]   (a tag means gpurun_out/prof_<tag>)"""
import csv, glob, os, sys
d = sys.argv[1] if os.path.isdir(sys.argv[1]) else f'gpurun_out/prof_{sys.argv[1]}'
f = max(glob.glob(f'{d}/*/*_kernel_stats.csv'), key=os.path.getmtime)
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{r['Name'][:58]:58s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:8.2f} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={float(r['Percentage']):5.1f}")
