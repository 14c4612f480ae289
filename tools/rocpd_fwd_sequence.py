"""Print every dispatch of the last complete forward pass in a rocprofv3 rocpd database: start offset, duration, gap to the
previous kernel's end, grid, short name.  A forward starts at input_to_rows_kernel.  Usage: rocpd_fwd_sequence.py db"""
import re
import sqlite3
import sys


def main(path, marker="input_to_rows_kernel"):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    dcols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    namecol = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    gx = "grid_size_x" if "grid_size_x" in dcols else ("grid_x" if "grid_x" in dcols else "0")
    wx = "workgroup_size_x" if "workgroup_size_x" in dcols else "1"
    rows = list(c.execute("select s.%s, d.start, d.end, d.%s, d.%s from %s d join %s s on d.kernel_id = s.id order by d.start"
                          % (namecol, gx, wx, kd, ks)))

    def short(n):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n)
        return re.sub(r"\(.*", "", n)[:90]
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    lo, hi = marks[-2], marks[-1]
    seg = rows[lo:hi]
    t0 = seg[0][1]
    print("forward span %.3f ms, %d dispatches, kernel time %.3f ms" % ((seg[-1][2] - t0) / 1e6, len(seg), sum(e - s for _, s, e, _, _ in seg) / 1e6))
    last = t0
    for name, s, e, g, w in seg:
        print("%8.1f %7.1f us  gap %5.1f  wg %6d  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - last) / 1e3, (g // w) if w else 0, short(name)))
        last = e


main(*sys.argv[1:])
