"""Memory-copy records of a rocprofv3 rocpd database: count and bytes by (direction, size).  Usage: rocpd_copies.py db"""
import collections
import sqlite3
import sys
c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
mc = [t for t in tabs if t.startswith("rocpd_memory_copy")]
print(mc)
for t in mc:
    cols = [r[1] for r in c.execute("pragma table_info(%s)" % t)]
    print(t, cols)
    rows = list(c.execute("select * from %s" % t))
    print(len(rows), "rows")
    szi = cols.index("size") if "size" in cols else None
    agg = collections.Counter()
    for r in rows:
        agg[(r[szi] if szi is not None else None)] += 1
    for k, v in agg.most_common(20):
        print("   size", k, "x", v)
    for r in rows[-12:]:
        print("   ", r)
