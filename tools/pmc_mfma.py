"""Per-kernel MFMA utilisation from one rocprofv3 PMC pass with SQ_VALU_MFMA_BUSY_CYCLES.
A dispatch has one counter row per shader-engine instance; they are summed.  The counter ticks once per cycle a SIMD's
MFMA pipe is busy (32 per v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md), so
    MfmaUtil = sum(busy cycles) / (kernel duration x clock x SIMDs),   clock 2.4 GHz, 1024 SIMDs (256 CUs x 4)
-- the gfx94x derived-counter formula with the duration taken from the dispatch timestamps (ROCm 7.2 ships no gfx950
derived-counter section).  Usage: pmc_mfma.py db [clock_ghz=2.4] [simds=1024]"""
import collections
import re
import sqlite3
import sys


def main(path, clock=2.4, simds=1024, top=14):
    clock, simds = float(clock), int(simds)
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
        agg[name][cn] = (n, tot)
    dur = {name: (n, t) for name, n, t in c.execute(
        "select s.%s, count(*), sum(d.end - d.start) from %s d join %s s on d.kernel_id = s.id group by s.%s"
        % (namecol, kd, ks, namecol))}
    rows = []
    for name, d in agg.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in d:
            continue
        n, mf = d["SQ_VALU_MFMA_BUSY_CYCLES"]
        nd, t = dur[name]
        rows.append((t, name, n, mf, t / nd))
    print("%-84s %7s %9s %18s %9s" % ("kernel", "calls", "avg_us", "mfma_busy_cyc/launch", "MfmaUtil%"))
    for t, name, n, mf, avg_ns in sorted(rows, reverse=True)[:top]:
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*", "", short)[:84]
        util = 100.0 * (mf / n) / (avg_ns * clock * simds)
        print("%-84s %7d %9.1f %18.0f %9.2f" % (short, n, avg_ns / 1e3, mf / n, util))


if __name__ == "__main__":
    main(*sys.argv[1:])
