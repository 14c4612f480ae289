"""Concurrency of the last full training step in a rocprofv3 rocpd database: per queue busy time, time with 0 / 1 / 2 / 3+ kernels
in flight, the largest all-idle gaps, and (per 1 ms bin) which queue was busy -- to see which stream is the critical path.
Usage: rocpd_overlap.py db [marker_kernel]"""
import collections
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
    marks = [i for i, r in enumerate(rows) if marker in r[0]]
    firsts = [m for j, m in enumerate(marks) if j == 0 or rows[m][1] - rows[marks[j - 1]][1] > 5e6]
    if len(firsts) < 3:
        print("not enough steps"); return
    seg = rows[firsts[-3]:firsts[-2]]
    t0, t1 = seg[0][1], max(r[2] for r in seg)
    span = (t1 - t0) / 1e6
    print("step span %.3f ms, %d dispatches" % (span, len(seg)))
    busy = collections.defaultdict(float)
    for n, s, e, q in seg:
        busy[q] += (e - s) / 1e6
    for q, b in sorted(busy.items(), key=lambda kv: -kv[1]):
        print("  queue %-4s busy %7.3f ms (%4.1f %% of the span), %d dispatches" % (q, b, 100 * b / span, sum(1 for r in seg if r[3] == q)))
    ev = []
    for n, s, e, q in seg:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    lvl, last, hist = 0, t0, collections.defaultdict(float)
    gaps = []
    for t, dlt in ev:
        hist[min(lvl, 3)] += (t - last) / 1e6
        if lvl == 0 and t - last > 0:
            gaps.append(((t - last) / 1e3, (last - t0) / 1e6))
        lvl += dlt; last = t
    print("  kernels in flight:  0: %.3f ms   1: %.3f ms   2: %.3f ms   3+: %.3f ms" % (hist[0], hist[1], hist[2], hist[3]))
    gaps.sort(reverse=True)
    print("  all-idle gaps: %d, total %.3f ms; largest: %s" % (len(gaps), sum(g for g, _ in gaps) / 1e3, ", ".join("%.0f us @ %.2f ms" % g for g in gaps[:8])))
    # time with exactly one kernel in flight, by queue and by kernel family
    solo_q, solo_k = collections.defaultdict(float), collections.defaultdict(float)
    active = {}
    ev2 = []
    for i, (n, s, e, q) in enumerate(seg):
        ev2.append((s, 1, i)); ev2.append((e, -1, i))
    ev2.sort()
    last = t0
    def fam(n):
        n = re.sub(r"\(anonymous namespace\)::", "", n)
        m = re.search(r"(conv_wgrad\w*|conv_igemm\w*|conv_m8\w*|conv_xjoin\w*|conv3x3_strip\w*|bn_\w+|stem_\w+|infonce\w*|sce\w*|sgd\w*|ema\w*|[A-Za-z_0-9]+)_kernel", n)
        return m.group(1) if m else n[:40]
    for t, dlt, i in ev2:
        if len(active) == 1:
            j = next(iter(active))
            solo_q[seg[j][3]] += (t - last) / 1e6
            solo_k[fam(seg[j][0])] += (t - last) / 1e6
        if dlt == 1: active[i] = 1
        else: active.pop(i, None)
        last = t
    print("  exactly one kernel in flight, by queue: " + ", ".join("q%s %.3f ms" % kv for kv in sorted(solo_q.items(), key=lambda kv: -kv[1])))
    print("  ... by kernel family: " + ", ".join("%s %.2f" % kv for kv in sorted(solo_k.items(), key=lambda kv: -kv[1])[:14]))
    # 1 ms bins
    qs = [q for q, _ in sorted(busy.items(), key=lambda kv: -kv[1])][:4]
    print("  per 1 ms bin, busy fraction of queues " + " ".join("q%s" % q for q in qs))
    nb = int(span) + 1
    for b in range(nb):
        lo, hi = t0 + b * 1e6, t0 + (b + 1) * 1e6
        fr = []
        for q in qs:
            tot = sum(max(0, min(e, hi) - max(s, lo)) for n, s, e, qq in seg if qq == q)
            fr.append(tot / 1e6)
        print("   %2d ms  " % b + "  ".join("%4.2f" % f for f in fr))


if __name__ == "__main__":
    main(*sys.argv[1:])
