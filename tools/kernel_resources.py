"""Per-kernel register / scratch / LDS / occupancy table of one translation unit (hipcc -Rpass-analysis=kernel-resource-usage).
Usage: python tools/kernel_resources.py vince_amd/csrc/conv_igemm_x3.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 else ""
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-result",
       "--cuda-device-only", "-c", src, "-o", "/dev/null", "-Rpass-analysis=kernel-resource-usage"] + sys.argv[3:]
err = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in err.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = {"name": subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip()}
        rows.append(cur)
        continue
    for key in ("VGPRs", "AGPRs", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "LDS Size [bytes/block]", "SGPRs"):
        m = re.search(re.escape(key) + r": (\d+)", line)
        if m and cur is not None:
            cur[key.split(" ")[0]] = int(m.group(1))
print("%-100s %5s %5s %7s %4s %7s" % ("kernel", "VGPR", "AGPR", "scratch", "occ", "LDS"))
for r in rows:
    n = re.sub(r"\(anonymous namespace\)::", "", r["name"])
    n = re.sub(r"\(vince_conv::ConvParams\)|\(vince_wgrad::WgradParams\)", "", n)
    if flt and flt not in n:
        continue
    print("%-100s %5d %5d %7d %4d %7d" % (n[:100], r.get("VGPRs", -1), r.get("AGPRs", -1), r.get("ScratchSize", -1), r.get("Occupancy", -1), r.get("LDS", -1)))
