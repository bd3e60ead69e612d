"""Per-kernel statistics from a rocprofv3 (rocpd, sqlite) kernel trace: name, calls, total, average."""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in cur.execute("pragma table_info(%s)" % ks)]
    name_col = "display_name" if "display_name" in cols else ("kernel_name" if "kernel_name" in cols else cols[-1])
    rows = cur.execute("select s.%s, count(*), sum(d.end - d.start), min(d.end - d.start), max(d.end - d.start) "
                       "from %s d join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (name_col, kd, ks, name_col)).fetchall()
    total = sum(r[2] for r in rows)
    print("name,calls,total_ms,avg_us,min_us,max_us,percent")
    for n, c, t, mn, mx in rows:
        n = re.sub(r"\(.*", "", n.replace("(anonymous namespace)::", "").replace("void ", ""))
        print("%s,%d,%.3f,%.2f,%.2f,%.2f,%.2f" % (n, c, t / 1e6, t / c / 1e3, mn / 1e3, mx / 1e3, 100.0 * t / total))


if __name__ == "__main__":
    main()
