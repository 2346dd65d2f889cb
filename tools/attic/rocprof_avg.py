"""Prints per-kernel call count and average duration from a rocprofv3 rocpd sqlite database (results.db)."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]; ks = [t for t in tabs if 'kernel_symbol' in t][0]
q = f"select s.kernel_name, count(*), avg(d.end-d.start) from {kd} d join {ks} s on d.kernel_id=s.id group by 1 order by 3 desc"
for r in c.execute(q): print(f'{r[0][:90]:90s} {r[1]:6d} {r[2] / 1e3:9.1f} us')
