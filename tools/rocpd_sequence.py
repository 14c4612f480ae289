"""Print the kernel sequence of the last full training step in a rocprofv3 rocpd database (run-length compressed,
with start offsets and gaps), to find stray launches and idle gaps.  Usage: rocpd_sequence.py db [marker_kernel]"""
import re
import sqlite3
import sys


def main(path, marker="sgd_kernel"):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    dcols = [r[1] for r in c.execute("pragma table_info(%s)" % kd)]
    namecol = "kernel_name" if "kernel_name" in scols else ("display_name" if "display_name" in scols else scols[-1])
    qcol = "queue_id" if "queue_id" in dcols else ("stream_id" if "stream_id" in dcols else "0")
    rows = list(c.execute("select s.%s, d.start, d.end, d.%s from %s d join %s s on d.kernel_id = s.id order by d.start"
                          % (namecol, qcol, kd, ks)))
    def short(n):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        n = re.sub(r"^void ", "", n)
        return re.sub(r"\(.*", "", n)[:70]
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    # a step ends with the trunk sgd launch; take the span between the 2nd-last pair of "first marker of a step"
    firsts = [m for j, m in enumerate(marks) if j == 0 or rows[m][1] - rows[marks[j - 1]][1] > 5e6]
    if len(firsts) < 3:
        print("not enough steps"); return
    lo, hi = firsts[-3], firsts[-2]
    seg = rows[lo:hi]
    t0 = seg[0][1]
    print("step span %.3f ms, %d dispatches" % ((seg[-1][2] - t0) / 1e6, len(seg)))
    prev_end = {}
    last_end = t0
    run = None
    for name, s, e, q in seg:
        gap = (s - last_end) / 1e3
        last_end = max(last_end, e)
        key = (short(name), q)
        if run and run[0] == key and gap < 20:
            run[1] += 1; run[2] += (e - s) / 1e3
            continue
        if run:
            print("%9.1f q%-3s x%-4d %9.1f us  %s" % (run[3], run[0][1], run[1], run[2], run[0][0]))
        if gap > 20:
            print("%9s ---- idle %.1f us" % ("", gap))
        run = [key, 1, (e - s) / 1e3, (s - t0) / 1e3]
    if run:
        print("%9.1f q%-3s x%-4d %9.1f us  %s" % (run[3], run[0][1], run[1], run[2], run[0][0]))


if __name__ == "__main__":
    main(*sys.argv[1:])
