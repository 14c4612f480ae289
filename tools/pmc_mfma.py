"""Per-kernel MFMA utilisation from one rocprofv3 PMC pass with SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CU_CYCLES and
GRBM_GUI_ACTIVE.  MfmaUtil follows the gfx94x derived-counter formula (ROCm ships no gfx950 section):
100 * SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE * CUs * 4).  Usage: pmc_mfma.py db [CUs=256]"""
import collections
import re
import sqlite3
import sys


def main(path, cus=256, top=14):
    cus = int(cus)
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    pi = [t for t in tabs if t.startswith("rocpd_info_pmc")][0]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    namecol = "kernel_name" if "kernel_name" in scols else "display_name"
    q = ("select s.%s, i.name, count(*), sum(e.value), sum(d.end-d.start) from %s e join %s i on e.pmc_id = i.id "
         "join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id group by s.%s, i.name"
         % (namecol, pe, pi, kd, ks, namecol))
    agg = collections.defaultdict(dict)
    for name, cn, n, tot, dur in c.execute(q):
        agg[name][cn] = (n, tot, dur)
    rows = []
    for name, d in agg.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in d or "GRBM_GUI_ACTIVE" not in d:
            continue
        n, mf, dur = d["SQ_VALU_MFMA_BUSY_CYCLES"]
        gui = d["GRBM_GUI_ACTIVE"][1]
        busy = d.get("SQ_BUSY_CU_CYCLES", (0, 0, 0))[1]
        rows.append((dur, name, n, mf, gui, busy))
    print("%-84s %6s %9s %16s %14s %9s" % ("kernel", "calls", "avg_us", "mfma_busy/launch", "gui_act/launch", "MfmaUtil%"))
    for dur, name, n, mf, gui, busy in sorted(rows, reverse=True)[:top]:
        short = re.sub(r"\(anonymous namespace\)::", "", name)
        short = re.sub(r"\(.*", "", short)[:84]
        util = 100.0 * mf / (gui * cus * 4) if gui else 0.0
        print("%-84s %6d %9.1f %16.0f %14.0f %9.2f" % (short, n, dur / n / 1e3, mf / n, gui / n, util))


if __name__ == "__main__":
    main(*sys.argv[1:])
