#!/usr/bin/env python3
"""Per-kernel calls / average / total from a rocprofv3 results.db (newer rocprofv3 writes a database, not kernel_stats.csv).
usage: scripts/kstats_db.py <dir-or-db> [top]"""
import glob, os, sqlite3, sys
p = sys.argv[1]
if os.path.isdir(p):
    p = max(glob.glob(os.path.join(p, "**", "*_results.db"), recursive=True), key=os.path.getmtime)
top = int(sys.argv[2]) if len(sys.argv) > 2 else 12
db = sqlite3.connect(p)
tabs = [r[0] for r in db.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
ks = [t for t in tabs if "kernel_symbol" in t][0]
rows = list(db.execute(f"select s.kernel_name, count(*), avg(d.end-d.start)/1e3, sum(d.end-d.start)/1e3 from {kd} d "
                       f"join {ks} s on d.kernel_id=s.id group by 1 order by 4 desc"))
total = sum(r[3] for r in rows)
for name, n, avg, tot in rows[:top]:
    print(f"{name[:70]:70s} calls {n:6d}  avg_us {avg:9.1f}  total_us {tot:10.0f}  {100*tot/total:5.1f}%")
