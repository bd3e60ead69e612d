"""top kernels of a rocprofv3 --kernel-trace database (rocpd sqlite).  usage: trace_top.py <results.db> [n]   (development aid)"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = db.execute("select name, count(*), sum(end-start)/1e6, avg(end-start)/1e3 from kernels group by name order by 3 desc limit %d" % n).fetchall()
for r in rows:
    print("%-92s %7d %9.2f ms %9.1f us" % (r[0][:92], r[1], r[2], r[3]))
