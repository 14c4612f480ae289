"""Summarise a rocprofv3 rocpd SQLite database (kernel-trace) into a per-kernel stats table (like --stats CSV)."""
import re
import sqlite3
import sys


def main(path, top=40):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    namecol = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    q = ("select s.%s, count(*), sum(d.end-d.start), min(d.end-d.start), max(d.end-d.start) from %s d join %s s "
         "on d.kernel_id = s.id group by s.%s order by 3 desc" % (namecol, kd, ks, namecol))
    rows = list(c.execute(q))
    total = sum(r[2] for r in rows)
    print("%-90s %8s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for name, n, tot, mn, mx in rows[:top]:
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*", "", short)[:90]
        print("%-90s %8d %12.1f %10.2f %10.2f %10.2f %6.2f" % (short, n, tot / 1e3, tot / n / 1e3, mn / 1e3, mx / 1e3, 100.0 * tot / total))
    print("TOTAL kernel time (us): %.1f over %d dispatches" % (total / 1e3, sum(r[1] for r in rows)))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 40)
