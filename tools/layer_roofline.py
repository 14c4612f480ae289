"""Per-layer roofline table from a VINCE_PROFILE_DUMP csv (one row per conv launch of the instrumented steps, each timed
alone between hipEvents): for every distinct layer shape -- algorithmic FLOPs and bytes, time, TF/s, TB/s, and which roof
binds it on this box (MFMA 2.5 PF dense bf16; HBM at the rate the library's own copy kernel reaches, default 5.28 TB/s).
usage: layer_roofline.py dump.csv steps [hbm_TBs] [x3]      (x3: the fp32-tagged launches ran as split-half products -- three 16-bit MFMAs
per algorithmic product -- and are priced against 2.5 PF / 3 instead of the fp32 MFMA peak)"""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
HBM = float(sys.argv[3]) if len(sys.argv) > 3 else 5.28
PEAK = 2500.0
F32_PEAK = PEAK / 3.0 if (len(sys.argv) > 4 and sys.argv[4] == "x3") else 157.3
names = ["ig %s/%s%s" % (t, sh, e) for t in ("f32", "bf16") for sh in ("64", "64x256", "128", "128x256") for e in ("", " bwd")] + ["wg f32", "wg bf16", "m8 bf16/256x256", "m8 bf16/256x256 bwd"]
agg = collections.OrderedDict()
for r in rows:
    if int(r["tag"]) >= len(names) or int(r["taps"]) == 0:     # streaming families (bytes, not conv shapes): bench.py's `kernels` has them
        continue
    key = (int(r["tag"]), int(r["M"]), int(r["Co"]), int(r["K"]), int(r["taps"]), int(r["stride"]), int(r["flags"]))
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += float(r["us"])
    a[2] += float(r["tflops"]) * float(r["us"])
print("%-18s %8s %5s %5s %4s %4s %6s %8s %8s %8s %7s %7s %7s %6s %s" % (
    "kernel", "M", "Co", "K", "taps", "s/os", "n/step", "GFLOP", "MB", "us", "TF/s", "TB/s", "%MFMA", "%HBM", "bound (roof time us)"))
tot = tot_roof = 0.0
for (tag, M, Co, K, taps, so, fl), (n, us, w) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    esz = 4 if (tag < 8 or tag == 16) else 2
    t_us = us / n
    tf = w / us
    gflop = tf * t_us * 1e-3        # TFLOP/s * us = MFLOP -> GFLOP
    sh = so // 10
    ci = K // taps
    wg = names[tag].startswith("wg")
    if wg:      # weight gradient: reads dY [M][Co] and the input [M * s^2][ci] once, writes fp32 [Co][K]
        nbytes = (M * Co + M * sh * sh * ci) * esz + Co * K * 4
    else:       # conv / dgrad: reads the input once (halo re-reads are L2 hits), writes the output, reads the weights
        nbytes = (M * sh * sh * ci + M * Co) * esz + Co * K * esz
        if "bwd" in names[tag] and (fl & 1):
            nbytes += M * Co * esz          # residual-gradient join: the old value of the output is read too
    t_mfma = gflop / PEAK * 1e3 if esz == 2 else gflop / F32_PEAK * 1e3       # us
    t_hbm = nbytes / (HBM * 1e6)                                           # us
    roof = max(t_mfma, t_hbm)
    tot += us / steps
    tot_roof += roof * n / steps
    print("%-18s %8d %5d %5d %4d %4d %6.1f %8.2f %8.1f %8.1f %7.1f %7.2f %6.1f%% %5.1f%% %s (%.1f)" % (
        names[tag], M, Co, K, taps, so, n / steps, gflop, nbytes / 1e6, t_us, tf, nbytes / t_us / 1e6,
        100 * tf / (PEAK if esz == 2 else F32_PEAK), 100 * nbytes / t_us / 1e6 / HBM, "hbm" if t_hbm >= t_mfma else "mfma", roof))
print("total %.2f ms/step measured, %.2f ms/step at the binding roof of every launch" % (tot / 1e3, tot_roof / 1e3))
