"""Per-dispatch value of one PMC counter in a rocprofv3 rocpd database, in launch order (kernels whose name contains `sub`).
Usage: pmc_dispatches.py db counter [sub]"""
import re
import sqlite3
import sys


def main(path, counter, sub=""):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    pi = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    namecol = "kernel_name" if "kernel_name" in scols else "display_name"
    q = ("select d.event_id, s.%s, d.start, d.end, sum(e.value) from %s e join %s i on e.pmc_id = i.id join %s d on e.event_id = d.event_id "
         "join %s s on d.kernel_id = s.id where i.name = ? group by d.event_id order by d.start" % (namecol, pe, pi, kd, ks))
    for ev, name, s, e, v in c.execute(q, (counter,)):
        if sub and sub not in name:
            continue
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*", "", short)[:70]
        print("%-72s %8.1f us %14.0f" % (short, (e - s) / 1e3, v))


main(*sys.argv[1:])
