"""Every PMC counter of a rocprofv3 rocpd database per kernel: value summed over the counter's instances, averaged per
dispatch, next to the average dispatch duration.  Usage: pmc_all.py db [name-substring]"""
import collections
import re
import sqlite3
import sys


def main(path, sub=""):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    pi = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    namecol = "kernel_name" if "kernel_name" in scols else "display_name"
    q = ("select s.%s, i.name, count(distinct e.event_id), sum(e.value) from %s e join %s i on e.pmc_id = i.id "
         "join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id group by s.%s, i.name"
         % (namecol, pe, pi, kd, ks, namecol))
    agg = collections.defaultdict(dict)
    for name, cn, n, tot in c.execute(q):
        agg[name][cn] = tot / max(n, 1)
    dur = {name: t / n for name, n, t in c.execute(
        "select s.%s, count(*), sum(d.end - d.start) from %s d join %s s on d.kernel_id = s.id group by s.%s"
        % (namecol, kd, ks, namecol))}
    for name, d in agg.items():
        if sub and sub not in name:
            continue
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*", "", short)[:100]
        print("%s  avg %.1f us" % (short, dur[name] / 1e3))
        for k in sorted(d):
            print("    %-34s %16.0f" % (k, d[k]))


main(*sys.argv[1:])
