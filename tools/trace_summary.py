#!/usr/bin/env python3
"""Per-kernel summary (count, total, average) of a rocprofv3 --kernel-trace results .db; optional divisor = steps."""
import sqlite3, sys
c = sqlite3.connect(sys.argv[1])
div = float(sys.argv[2]) if len(sys.argv) > 2 else 1.0
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
kd = [t for t in tabs if 'kernel_dispatch' in t][0]
ks = [t for t in tabs if 'kernel_symbol' in t][0]
rows = c.execute(f"select s.kernel_name, count(*), sum(d.end-d.start), avg(d.end-d.start) from {kd} d join {ks} s "
                 f"on d.kernel_id=s.id group by s.kernel_name order by 3 desc").fetchall()
tot = sum(r[2] for r in rows); n = sum(r[1] for r in rows)
print("total %.3f ms, %d launches  (per step: %.3f ms, %.0f launches)" % (tot / 1e6, n, tot / 1e6 / div, n / div))
for r in rows[:int(sys.argv[3]) if len(sys.argv) > 3 else 30]:
    print("%-100s %7d %10.1f us %9.2f us avg %5.1f%%" % (r[0][:100], r[1], r[2] / 1e3, r[3] / 1e3, 100 * r[2] / tot))
