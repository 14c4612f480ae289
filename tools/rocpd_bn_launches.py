"""Per-launch list (grid, duration) of the BatchNorm kernels in a rocprofv3 rocpd database, grouped by (kernel, grid):
which BatchNorm launches run below the streaming rate.  Usage: rocpd_bn_launches.py db"""
import re
import sqlite3
import sys
from collections import defaultdict


def main(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    dcols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    namecol = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    gx = [x for x in dcols if "grid" in x.lower()]
    q = "select s.%s, d.end - d.start, %s from %s d join %s s on d.kernel_id = s.id" % (
        namecol, ", ".join("d." + g for g in gx), kd, ks)
    groups = defaultdict(list)
    for row in c.execute(q):
        name = row[0]
        if "bn_" not in name and "stem_" not in name:
            continue
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"<.*", "", re.sub(r"^void ", "", short))[:24]
        groups[(short, tuple(row[2:]))].append(row[1] / 1e3)
    print("grid columns:", gx)
    for (name, grid), v in sorted(groups.items(), key=lambda kv: -sum(kv[1])):
        print("%-24s grid %-22s n %5d  avg %8.1f us  total %9.1f us" % (name, grid, len(v), sum(v) / len(v), sum(v)))


if __name__ == "__main__":
    main(sys.argv[1])
