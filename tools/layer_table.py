"""Aggregate a VINCE_PROFILE_DUMP csv (per-launch conv timings) by layer shape."""
import collections
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1
agg = collections.OrderedDict()
for r in rows:
    key = (r["tag"], r["M"], r["Co"], r["K"], r["taps"], r["stride"], r["flags"])
    a = agg.setdefault(key, [0, 0.0, 0.0])
    a[0] += 1
    a[1] += float(r["us"])
    a[2] += float(r["tflops"]) * float(r["us"])
names = ["ig %s/%s%s" % (t, sh, e) for t in ("f32", "bf16") for sh in ("64", "64x256", "128", "128x256") for e in ("", " bwd")] + ["wg f32", "wg bf16"]
print("%-20s %9s %6s %6s %4s %4s %3s %6s %9s %8s %8s" % ("kernel", "M", "Co", "K", "taps", "s/os", "fl", "n/step", "us/launch", "ms/step", "TF/s"))
tot = 0
for k, (n, us, w) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    tot += us
    print("%-20s %9s %6s %6s %4s %4s %3s %6.1f %9.1f %8.3f %8.1f" % (names[int(k[0])], k[1], k[2], k[3], k[4], k[5], k[6], n / steps, us / n, us / steps / 1e3, w / us))
print("total ms/step %.2f" % (tot / steps / 1e3))
