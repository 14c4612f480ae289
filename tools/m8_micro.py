"""conv_m8 (8-wavefront 256 x 256 core) against conv_igemm's own tiles, layer by layer at the benchmark batch.

    VINCE_KNOBS=m8=0 python tools/m8_micro.py save /tmp/m8ref     # reference outputs + timings of the 4-wavefront tiles
    VINCE_KNOBS=m8_min_k=1,m8_min_tiles=1 python tools/m8_micro.py check /tmp/m8ref    # the same launches through conv_m8: bitwise comparison + timings

Each shape runs `reps` times and every repetition is compared (a race in the LDS-DMA pipeline shows up as a rare mismatch)."""
import os
import sys
import torch
sys.path.insert(0, ".")
from vince_amd import ops  # noqa: E402

mode, path = sys.argv[1], sys.argv[2]
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 30
os.makedirs(path, exist_ok=True)
N = int(os.environ.get("M8_BATCH", "256"))
# hw, ci, co, k, stride
SHAPES = [(14, 256, 256, 3, 1), (14, 1024, 256, 1, 1), (14, 256, 1024, 1, 1), (7, 512, 512, 3, 1), (7, 2048, 512, 1, 1),
          (7, 512, 2048, 1, 1), (14, 1024, 512, 1, 1), (28, 512, 256, 1, 1), (14, 1024, 2048, 1, 2), (28, 256, 256, 3, 2),
          (28, 128, 512, 1, 1), (56, 64, 256, 1, 1)]
g = torch.Generator(device="cuda").manual_seed(7)
print("%-28s %9s %9s %9s  %s" % ("shape", "us", "TF/s", "TB/s", "check"))
for hw, ci, co, k, st in SHAPES:
    x = torch.randn(N, hw, hw, ci, device="cuda", generator=g).clamp_(min=0).bfloat16()
    w = (torch.randn(co, k * k, ci, device="cuda", generator=g) * (2.0 / (ci * k * k)) ** 0.5).bfloat16()
    d = ops.conv_desc(N, hw, hw, ci, co, k, st, k // 2)
    out = torch.empty(N, d.Ho, d.Wo, co, device="cuda", dtype=torch.bfloat16)
    stats = torch.zeros(ops.STATS_REPLICAS, co, 2, device="cuda", dtype=torch.float64)
    name = "%dx%d %d->%d k%d s%d" % (hw, hw, ci, co, k, st)
    f = os.path.join(path, name.replace(" ", "_").replace(">", "") + ".pt")
    ref = torch.load(f) if mode == "check" else None
    bad = 0
    for r in range(reps):
        out.zero_()
        stats.zero_()
        ops.conv_igemm(d, x, w, out, stats=stats)
        if ref is not None:
            if not torch.equal(out.view(torch.int16), ref["out"].cuda().view(torch.int16)):
                bad += 1
                if bad == 1:
                    diff = (out.float() - ref["out"].cuda().float()).abs()
                    idx = diff.flatten().argmax().item()
                    print("   first mismatch rep %d: max diff %.4g at flat index %d (pixel %d, channel %d), %d elements differ" %
                          (r, diff.max().item(), idx, idx // co, idx % co, int((diff > 0).sum())))
    torch.cuda.synchronize()
    if mode == "save":
        torch.save({"out": out.cpu(), "stats": stats.sum(0).cpu()}, f)
        chk = "saved"
    else:
        srel = ((stats.sum(0).cpu() - ref["stats"]).abs() / (ref["stats"].abs() + 1e-9)).max().item()
        chk = "bitwise %s (%d/%d reps differ), stats rel %.1e" % ("OK" if bad == 0 else "MISMATCH", bad, reps, srel)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    n = 30
    e0.record()
    for _ in range(n):
        ops.conv_igemm(d, x, w, out, stats=stats)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1000 / n
    M = N * d.Ho * d.Wo
    fl = 2.0 * M * co * ci * k * k
    by = (x.numel() + out.numel() + w.numel()) * 2
    print("%-28s %9.1f %9.1f %9.2f  %s" % (name, us, fl / us / 1e6, by / us / 1e6, chk))
