import csv, glob, sys
f = sorted(glob.glob(f'gpurun_out/prof_{sys.argv[1]}/*/*_kernel_stats.csv'))[-1]
for r in list(csv.DictReader(open(f)))[:int(sys.argv[2]) if len(sys.argv) > 2 else 20]:
    print(f"{r['Name'][:58]:58s} calls={r['Calls']:>6s} total_ms={float(r['TotalDurationNs'])/1e6:8.2f} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={float(r['Percentage']):5.1f}")
