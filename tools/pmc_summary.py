"""Combine the FETCH_SIZE and WRITE_SIZE rocprofv3 PMC passes (two rocpd databases) into per-kernel HBM bytes per
launch, keyed by bench.py's kernel tags.  FETCH_SIZE is doubled (gfx950 tallies 128-byte requests at 64 bytes,
MI355X_MICROARCH.md HBM section); both counters are in KB.  Usage: pmc_summary.py fetch.db write.db out.json "source" """
import json
import re
import sqlite3
import sys


def tag_of(name):
    # families without a tile-shape split: bench.py's KERNEL_TAGS names
    m8 = re.search(r"conv_m8_kernel(?:ILi|<)([012])", name)
    if m8:
        return "conv_m8<bf16,256ch x 256px,%s>" % ("bwd" if m8.group(1) == "1" else "fwd")
    xj = re.search(r"conv_xjoin_kernel(?:ILi\d+ELi\d+ELb([01])ELb([01])ELb([01])ELb([01])E|<\d+, \d+, (true|false), (true|false), (true|false), (true|false)>)", name)
    if xj:
        g = [v for v in xj.groups() if v is not None]
        plain, dgrad = g[2] in ("1", "true"), g[3] in ("1", "true")
        return "conv_xjoin<dgrad>" if dgrad else ("conv_xjoin<stats>" if plain else "conv_xjoin<join>")
    if "conv3x3_strip_kernel" in name:
        return "conv3x3_strip"
    if "bn_bwd_apply_kernel" in name:
        return "bn_bwd_apply"
    if "bn_bwd_reduce_kernel" in name:
        return "bn_bwd_reduce"
    if "bn_apply_kernel" in name or "bn_apply_gram_kernel" in name:
        return "bn_apply"
    if "stem_pool_fwd_kernel" in name:
        return "stem_pool_fwd"
    if "stem_bwd_kernel" in name or "stem_pool_bwd_kernel" in name:
        return "stem_bwd"
    x3 = re.search(r"conv_igemm_dlds_kernelI5x3([hb])_tLi(\d+)ELi\d+ELi\d+ELi\d+ELi(\d+)EL[bi]([012])E", name)
    if x3:      # the split-half element types (compute_dtype "x3"): fp32 tensors, IEEE-half (h) / bfloat16 (b) halves
        return "conv_igemm<x3%s,%sch x %spx,%s>" % (x3.group(1), x3.group(2), x3.group(3), "bwd" if x3.group(4) in ("1", "2") else "fwd")
    if "conv_wgrad_x3_kernel" in name:
        return "conv_wgrad<x3b>"
    m = re.search(r"conv_igemm_dlds_kernelI([tf])Li(\d+)ELi\d+ELi\d+ELi\d+ELi(\d+)EL[bi]([012])E", name)
    if not m:
        m2 = re.search(r"conv_igemm_dlds_kernel<(unsigned short|float), (\d+), \d+, \d+, \d+, (\d+), (true|false|0|1|2)[,>]", name)
        if not m2:
            if "conv_wgrad_tr" in name:     # the transposing-read kernel is bf16 only (no type parameter in its name)
                return "conv_wgrad<bf16>"
            if "conv_wgrad" in name:
                return "conv_wgrad<%s>" % ("bf16" if ("It" in name or "unsigned short" in name) else "f32")
            return None
        t, ct, ptl, b = m2.group(1) == "unsigned short", int(m2.group(2)), int(m2.group(3)), m2.group(4) in ("true", "1", "2")
    else:
        t, ct, ptl, b = m.group(1) == "t", int(m.group(2)), int(m.group(3)), m.group(4) in ("1", "2")
    return "conv_igemm<%s,%dch x %dpx,%s>" % ("bf16" if t else "f32", ct, ptl, "bwd" if b else "fwd")


def per_kernel(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    namecol = "kernel_name" if "kernel_name" in scols else "display_name"
    q = ("select s.%s, count(*), sum(e.value) from %s e join %s d on e.event_id = d.event_id "
         "join %s s on d.kernel_id = s.id group by s.%s" % (namecol, pe, kd, ks, namecol))
    out = {}
    for name, n, tot in c.execute(q):
        t = tag_of(name)
        if t:
            a = out.setdefault(t, [0, 0.0])
            a[0] += n
            a[1] += tot
    return out


def step_total(path):
    """Counter total (KB) over the dispatches of the last complete training step (between the first sgd_kernel launch
    of two consecutive steps)."""
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    namecol = "kernel_name" if "kernel_name" in scols else "display_name"
    rows = list(c.execute("select s.%s, d.start, e.value from %s e join %s d on e.event_id = d.event_id join %s s "
                          "on d.kernel_id = s.id order by d.start" % (namecol, pe, kd, ks)))
    marks = [i for i, r in enumerate(rows) if "sgd_kernel" in r[0]]
    firsts = [m for j, m in enumerate(marks) if j == 0 or rows[m][1] - rows[marks[j - 1]][1] > 5e6]
    if len(firsts) < 2:
        return None
    lo, hi = firsts[-2], firsts[-1]
    return sum(r[2] for r in rows[lo:hi])


def main(fetch_db, write_db, out_json, source):
    f, w = per_kernel(fetch_db), per_kernel(write_db)
    sf, sw = step_total(fetch_db), step_total(write_db)
    kernels = {}
    for t in sorted(set(f) & set(w)):
        n = f[t][0]
        fetch_b, write_b = 2.0 * f[t][1] * 1024.0 / n, w[t][1] * 1024.0 / w[t][0]
        kernels[t] = {"launches": n, "fetch_bytes_per_launch": int(fetch_b), "write_bytes_per_launch": int(write_b),
                      "bytes_per_launch": int(fetch_b + write_b)}
    step = None
    if sf is not None and sw is not None:
        step = {"fetch_bytes": int(2.0 * sf * 1024.0), "write_bytes": int(sw * 1024.0),
                "bytes": int(2.0 * sf * 1024.0 + sw * 1024.0)}
        print("whole training step: fetch %.2f GB  write %.2f GB  total %.2f GB" % (step["fetch_bytes"] / 1e9,
              step["write_bytes"] / 1e9, step["bytes"] / 1e9))
    # stamp: which sources these counters were taken from (bench.py prints traffic_stale when the built kernels differ)
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    try:
        from bench import kernel_source_hash
        khash = kernel_source_hash()
    except Exception:
        khash = None
    head = os.environ.get("VINCE_GIT_HEAD")      # the GPU box has no .git: the caller exports it (tools/final_profile.sh)
    if not head:
        try:
            head = subprocess.run(["git", "-C", root, "rev-parse", "--short=12", "HEAD"], capture_output=True, text=True).stdout.strip() or None
        except Exception:
            head = None
    json.dump({"kernels": kernels, "step": step, "source": source, "kernel_source_hash": khash, "git_head": head},
              open(out_json, "w"), indent=1)
    for t, v in kernels.items():
        print("%-40s launches %5d  fetch %8.1f MB  write %8.1f MB  total %8.1f MB / launch"
              % (t, v["launches"], v["fetch_bytes_per_launch"] / 1e6, v["write_bytes_per_launch"] / 1e6, v["bytes_per_launch"] / 1e6))


if __name__ == "__main__":
    main(*sys.argv[1:5])
