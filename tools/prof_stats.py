"""Per-kernel durations out of a rocprofv3 results .db (the default output format): name, launches, average and minimum in us.
usage: prof_stats.py results.db [substring]"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if "kernel_dispatch" in t][0]
sym = [t for t in tabs if "info_kernel_symbol" in t][0]
want = sys.argv[2] if len(sys.argv) > 2 else ""
q = f"select s.kernel_name, count(*), avg(d.end-d.start), min(d.end-d.start) from {kd} d join {sym} s on d.kernel_id=s.id group by s.kernel_name order by 3 desc"
for name, n, avg, mn in c.execute(q):
    if want in name:
        print(f"{name[:90]:90s} n={n:5d} avg={avg / 1e3:9.1f} us min={mn / 1e3:9.1f} us")
