"""Per-kernel average of one PMC counter (FETCH_SIZE / WRITE_SIZE, kilobytes) from a rocprofv3 rocpd database."""
import re
import sqlite3
import sys


def main(path, top=12):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    pi = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    namecol = "kernel_name" if "kernel_name" in scols else "display_name"
    cname = list(c.execute("select name from %s limit 1" % pi))[0][0]
    q = ("select s.%s, count(*), sum(e.value), sum(d.end-d.start) from %s e join %s d on e.event_id = d.event_id "
         "join %s s on d.kernel_id = s.id group by s.%s order by 3 desc" % (namecol, pe, kd, ks, namecol))
    print("counter:", cname, "(KB)")
    print("%-80s %7s %14s %12s %10s" % ("kernel", "calls", "total_MB", "avg_MB", "avg_us"))
    for name, n, tot, dur in list(c.execute(q))[:top]:
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*", "", short)[:80]
        print("%-80s %7d %14.1f %12.3f %10.1f" % (short, n, tot / 1024.0, tot / 1024.0 / n, dur / n / 1e3))


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 12)
