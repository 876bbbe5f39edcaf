"""Summarise a rocprofv3 rocpd sqlite (--kernel-trace --stats) into a per-kernel table (text)."""
import sqlite3, sys, re
db = sqlite3.connect(sys.argv[1])
c = db.cursor()
rows = c.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels group by name order by sum(duration) desc").fetchall()
tot = sum(r[2] for r in rows)
span = c.execute("select min(start), max(end) from kernels").fetchone()
print("# rocprofv3 --kernel-trace --stats summary: %s" % sys.argv[1])
print("# total kernel time %.3f ms over %d dispatches; first-to-last span %.3f ms" % (tot / 1e6, sum(r[1] for r in rows), (span[1] - span[0]) / 1e6))
print("%-90s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_ms", "avg_us", "min_us", "max_us", "pct"))
for r in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    name = re.sub(r"\s+", " ", r[0])[:90]
    print("%-90s %7d %12.3f %10.2f %10.2f %10.2f %6.2f" % (name, r[1], r[2] / 1e6, r[3] / 1e3, r[4] / 1e3, r[5] / 1e3, 100.0 * r[2] / tot))
