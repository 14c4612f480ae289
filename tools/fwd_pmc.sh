#!/bin/bash
# GPU box: HBM bytes of ONE no-grad train-mode forward (ResNet-50, B=256, bf16 -- the trunk of the forward + InfoNCE leg) from the PMC
# counters, separate passes as MI355X_MICROARCH.md prescribes: FETCH_SIZE (x2 on gfx950: 128-byte requests tallied at 64 B) + WRITE_SIZE.
# Usage: bash tools/fwd_pmc.sh [out dir]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=${1:-gpurun_out/fwd_pmc}; rm -rf $O; mkdir -p $O
N=6
for C in FETCH_SIZE WRITE_SIZE; do
  timeout 900 rocprofv3 --kernel-trace --pmc $C -d $O/pmc_$C -o pmc -- python tools/fwd_profile.py $N > $O/pmc_$C.log 2>&1
done
F=$(find $O/pmc_FETCH_SIZE -name '*.db' | head -1); W=$(find $O/pmc_WRITE_SIZE -name '*.db' | head -1)
python - "$F" "$W" $N <<'PY' > $O/fwd_pmc_summary.txt
import re, sqlite3, sys
def per_kernel(path):
    c = sqlite3.connect(path)
    tabs = [r[0] for r in c.execute("select name from sqlite_master where type='table'")]
    kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
    ks = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
    pe = [t for t in tabs if t.startswith("rocpd_pmc_event")][0]
    scols = [r[1] for r in c.execute("pragma table_info(%s)" % ks)]
    namecol = "kernel_name" if "kernel_name" in scols else "display_name"
    q = ("select s.%s, count(*), sum(e.value) from %s e join %s d on e.event_id = d.event_id join %s s on d.kernel_id = s.id group by s.%s"
         % (namecol, pe, kd, ks, namecol))
    return {n: (k, v) for n, k, v in c.execute(q)}
f, w, nfwd = per_kernel(sys.argv[1]), per_kernel(sys.argv[2]), int(sys.argv[3]) + 3      # + the 3 warm-up forwards of tools/fwd_profile.py
rows = []
for name in set(f) | set(w):
    fk, fv = f.get(name, (0, 0)); wk, wv = w.get(name, (0, 0))
    rows.append((2.0 * fv / 1024 + wv / 1024, name, max(fk, wk), 2.0 * fv / 1024, wv / 1024))   # MB; FETCH_SIZE x2 (gfx950 correction)
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("HBM bytes of the no-grad train-mode forward, ResNet-50 B=256 bf16 (PMC: 2 x FETCH_SIZE + WRITE_SIZE, separate passes; %d forwards traced)" % nfwd)
print("per forward: %.2f GB  (fetch %.2f + write %.2f)" % (tot / 1024 / nfwd, sum(r[3] for r in rows) / 1024 / nfwd, sum(r[4] for r in rows) / 1024 / nfwd))
print("%-86s %7s %12s %12s %12s" % ("kernel", "calls", "MB/forward", "fetch MB/fw", "write MB/fw"))
for t, name, k, fv, wv in rows[:28]:
    short = re.sub(r"\(anonymous namespace\)::", "", name); short = re.sub(r"\(.*", "", short)[:86]
    print("%-86s %7d %12.1f %12.1f %12.1f" % (short, k, t / nfwd, fv / nfwd, wv / nfwd))
PY
rm -rf $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
cat $O/fwd_pmc_summary.txt | head -12
